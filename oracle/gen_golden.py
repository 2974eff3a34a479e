#!/usr/bin/env python
"""oracle/gen_golden.py -- TEST INFRASTRUCTURE: captures golden vectors from the REFERENCE.

Run in the build container only (needs /root/reference):

    python -m oracle.gen_golden            # writes tests/golden/*.npz

It imports the reference's own gnngls.operators / gnngls.algorithms / gnngls.models (see
oracle/ref_import.py for the two stubs this needs), runs them on seeded synthetic inputs and
stores inputs + outputs as small .npz fixtures (fp64 stored as fp64: bit exact).  The fixtures
are data only; no reference source travels.  tests/test_oracle_golden.py pins oracle/ against
them, tests/test_gls_gpu.py / test_model_gpu.py pin the HIP path against them.

Golden groups:
  ops_n{n}.npz        two_opt_cost / relocate_cost full tables, a2a / o2a results (both
                      first_improvement settings)                          operators.py:6-147
  ops_ties.npz        integer-valued D with ~1e-8 noise: exact ties + np.isclose(0, delta) edge
  ls_n{n}.npz         local_search final tour / cost / cost trace          algorithms.py:111-132
  gls_*.npz           guided_local_search under a fake clock (exactly K outer iterations):
                      trace, best tour/cost, final penalties              algorithms.py:135-195
  misc.npz            nearest_neighbor (zero ties), tour_cost, optimal_cost, MinMax scalers
  model_n{n}.npz      EdgePropertyPredictionModel forward (reference wiring, GATConv shim)
"""
import itertools
import os
import sys

import networkx as nx
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def make_graph(pos):
    """scripts/generate_instances.py:25-33 without the Concorde call."""
    G = nx.Graph()
    for n, p in enumerate(pos):
        G.add_node(n, pos=p)
    for i, j in itertools.combinations(G.nodes, 2):
        w = np.linalg.norm(G.nodes[j]["pos"] - G.nodes[i]["pos"])
        G.add_edge(i, j, weight=w)
    return G


def graph_from_matrix(D):
    n = D.shape[0]
    G = nx.Graph()
    G.add_nodes_from(range(n))
    for i, j in itertools.combinations(range(n), 2):
        G.add_edge(i, j, weight=np.float64(D[i, j]))
    return G


def random_tour(rng, n):
    return [0] + (1 + rng.permutation(n - 1)).tolist() + [0]


def capture_ops(ref, tour, D):
    ops = ref.operators
    n = len(tour) - 1
    two = np.full((n + 1, n + 1), np.nan)
    rel = np.full((n + 1, n + 1), np.nan)
    for i in range(1, n):
        for j in range(1, n):
            two[i, j] = ops.two_opt_cost(tour, D, i, j)
            rel[i, j] = ops.relocate_cost(tour, D, i, j)
    out = dict(tour=np.array(tour, dtype=np.int32), D=np.asarray(D, dtype=np.float64),
               two_opt_table=two, relocate_table=rel)
    for fi in (0, 1):
        for name in ("two_opt_a2a", "relocate_a2a"):
            d, t = getattr(ops, name)(tour, D, bool(fi))
            out[f"{name}_fi{fi}_delta"] = np.float64(d)
            out[f"{name}_fi{fi}_tour"] = np.array(t, dtype=np.int32)
        for name in ("two_opt_o2a", "relocate_o2a"):
            ds, ts = [], []
            for i in range(1, n):
                d, t = getattr(ops, name)(tour, D, i, bool(fi))
                ds.append(d)
                ts.append(t)
            out[f"{name}_fi{fi}_delta"] = np.array(ds, dtype=np.float64)
            out[f"{name}_fi{fi}_tour"] = np.array(ts, dtype=np.int32)
    return out


class FakeClock:
    """Replaces the `time` module inside gnngls.algorithms: time() = completed local_search
    calls - 1, so `while time.time() < t_lim` (algorithms.py:146) runs exactly t_lim outer
    iterations."""

    def __init__(self):
        self.ls_calls = 0

    def time(self):
        return self.ls_calls - 1


def run_ref_gls(ref, G, init_tour, init_cost, K, guides, perturbation_moves, first_improvement):
    alg = ref.algorithms
    clock = FakeClock()
    real_time, real_ls = alg.time, alg.local_search

    def counted_ls(*a, **kw):
        r = real_ls(*a, **kw)
        clock.ls_calls += 1
        return r

    alg.time, alg.local_search = clock, counted_ls
    try:
        best_tour, best_cost, progress = alg.guided_local_search(
            G, init_tour, init_cost, K, weight="weight", guides=guides,
            perturbation_moves=perturbation_moves, first_improvement=first_improvement)
    finally:
        alg.time, alg.local_search = real_time, real_ls
    pen, _ = nx.attr_matrix(G, "penalty")
    return best_tour, best_cost, [r["cost"] for r in progress], np.asarray(pen)


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = ref_import.import_reference(with_models=True)
    rng = np.random.default_rng(20211011)

    # ---- (1),(2) operators ------------------------------------------------------------------
    for n in (5, 8, 20, 50, 100):
        G = make_graph(rng.random((n, 2)))
        D, _ = nx.attr_matrix(G, "weight")
        D = np.asarray(D)
        np.savez_compressed(os.path.join(GOLD, f"ops_n{n}.npz"), **capture_ops(ref, random_tour(rng, n), D))

    # crafted ties / isclose edge: integer lattice distances + ~1e-8 symmetric noise
    n = 12
    base = rng.integers(1, 4, size=(n, n)).astype(np.float64)
    base = np.triu(base, 1)
    base = base + base.T
    noise = np.triu(rng.choice([0.0, 5e-9, 1e-8, 1.00001e-8, 1.0001e-8, 2e-8, -5e-9, -1e-8], size=(n, n)), 1)
    Dt = base + noise + noise.T
    cases = {}
    for c in range(6):
        t = random_tour(rng, n)
        for k, v in capture_ops(ref, t, Dt if c % 2 else base).items():
            cases[f"c{c}_{k}"] = v
    np.savez_compressed(os.path.join(GOLD, "ops_ties.npz"), n_cases=6, **cases)

    # ---- (3) local_search ---------------------------------------------------------------------
    for n in (8, 20, 50, 100):
        G = make_graph(rng.random((n, 2)))
        D = np.asarray(nx.attr_matrix(G, "weight")[0])
        out = dict(D=D)
        for fi in (0, 1):
            t0 = random_tour(rng, n)
            c0 = ref.tour_cost(G, t0)
            t, c, prog = ref.algorithms.local_search(t0, c0, D, bool(fi))
            out[f"fi{fi}_init_tour"] = np.array(t0, dtype=np.int32)
            out[f"fi{fi}_init_cost"] = np.float64(c0)
            out[f"fi{fi}_tour"] = np.array(t, dtype=np.int32)
            out[f"fi{fi}_cost"] = np.float64(c)
            out[f"fi{fi}_trace"] = np.array([r["cost"] for r in prog], dtype=np.float64)
        np.savez_compressed(os.path.join(GOLD, f"ls_n{n}.npz"), **out)

    # ---- (4) guided_local_search under a fake clock -------------------------------------------
    gls_cases = [
        # n, K, guides, perturbation_moves, first_improvement
        (10, 8, ["weight"], 5, False),
        (20, 5, ["regret_pred"], 20, False),
        (20, 20, ["regret_pred", "weight"], 20, False),
        (20, 10, ["weight"], 30, True),
        (50, 5, ["regret_pred"], 20, False),
        (50, 20, ["weight", "regret_pred"], 20, False),
        (100, 5, ["regret_pred"], 20, False),
        (100, 20, ["regret_pred"], 20, False),
        (100, 5, ["weight"], 20, True),
    ]
    for ci, (n, K, guides, pm, fi) in enumerate(gls_cases):
        G = make_graph(rng.random((n, 2)))
        # synthetic regret predictions: float32 values clamped at 0 (scripts/test.py:79-83) -> many exact 0 ties
        for e in G.edges:
            G.edges[e]["regret_pred"] = np.maximum(np.float32(rng.normal(0.05, 0.1)).item(), 0)
        D = np.asarray(nx.attr_matrix(G, "weight")[0])
        W = {g: np.asarray(nx.attr_matrix(G, g)[0]) for g in guides}
        init_guide = guides[0]
        init_tour = ref.algorithms.nearest_neighbor(G, 0, weight=init_guide)      # test.py:85
        init_cost = ref.tour_cost(G, init_tour)                                  # test.py:90
        best_tour, best_cost, trace, pen = run_ref_gls(ref, G, list(init_tour), init_cost, K, guides, pm, fi)
        # prefix property: the K-iteration trace is a prefix of the 2K-iteration trace
        nx.set_edge_attributes(G, 0, "penalty")
        _, _, trace2, _ = run_ref_gls(ref, G, list(init_tour), init_cost, K + 3, guides, pm, fi)
        assert trace2[:len(trace)] == trace
        assert ref.is_valid_tour(G, best_tour)
        np.savez_compressed(
            os.path.join(GOLD, f"gls_c{ci}_n{n}_K{K}.npz"),
            D=D, guides=np.stack([W[g] for g in guides]), guide_names=np.array(guides),
            init_tour=np.array(init_tour, dtype=np.int32), init_cost=np.float64(init_cost),
            K=K, perturbation_moves=pm, first_improvement=int(fi),
            best_tour=np.array(best_tour, dtype=np.int32), best_cost=np.float64(best_cost),
            trace=np.array(trace, dtype=np.float64), penalty=pen.astype(np.int32))

    # ---- (5),(6),(8) misc ---------------------------------------------------------------------
    misc = {}
    n = 30
    G = make_graph(rng.random((n, 2)))
    for e in G.edges:
        G.edges[e]["regret_pred"] = np.maximum(np.float32(rng.normal(-0.05, 0.1)).item(), 0)   # mostly zeros
        G.edges[e]["in_solution"] = bool(rng.random() < 0.1)
    misc["nn_W_weight"] = np.asarray(nx.attr_matrix(G, "weight")[0])
    misc["nn_W_regret"] = np.asarray(nx.attr_matrix(G, "regret_pred")[0])
    misc["nn_tour_weight"] = np.array(ref.algorithms.nearest_neighbor(G, 0, weight="weight"), dtype=np.int32)
    misc["nn_tour_regret"] = np.array(ref.algorithms.nearest_neighbor(G, 0, weight="regret_pred"), dtype=np.int32)
    t = random_tour(rng, n)
    misc["tc_tour"] = np.array(t, dtype=np.int32)
    misc["tc_cost"] = np.float64(ref.tour_cost(G, t))
    misc["opt_in_solution"] = np.asarray(nx.attr_matrix(G, "in_solution")[0]).astype(np.int8)
    misc["opt_cost"] = np.float64(ref.optimal_cost(G))
    # MinMaxScaler arithmetic on float32 (datasets.py:84-89, test.py:79) via sklearn itself
    from sklearn.preprocessing import MinMaxScaler
    w32 = np.array([[G.edges[e]["weight"]] for e in G.edges], dtype=np.float32)
    sc = MinMaxScaler().fit(np.vstack([w32, np.float32([[1.3]]), np.float32([[0.01]])]))
    misc["scaler_scale"] = sc.scale_.astype(np.float64)
    misc["scaler_min"] = sc.min_.astype(np.float64)
    misc["scaler_in"] = w32
    misc["scaler_fwd"] = sc.transform(w32)
    y32 = rng.normal(0.3, 0.4, size=w32.shape).astype(np.float32)
    misc["scaler_inv_in"] = y32
    misc["scaler_inv"] = sc.inverse_transform(y32)
    assert misc["scaler_fwd"].dtype == np.float32 and misc["scaler_inv"].dtype == np.float32
    np.savez_compressed(os.path.join(GOLD, "misc.npz"), **misc)

    # ---- (7) model forward through the reference's own wiring ---------------------------------
    import torch
    from oracle import model_oracle as mo

    torch.manual_seed(1234)
    ref_model = ref.models.EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
    sd = mo.synthetic_state_dict(ref_model, seed=99)
    ref_model.load_state_dict(sd)
    ref_model.eval()
    assert len(ref_model.message_passing_layers) == 8          # models.py:59-61 (n_heads layers)
    checksum = float(sum(v.double().abs().sum() for k, v in sd.items() if v.dtype.is_floating_point))
    for n in (5, 10, 20):
        G = mo.line_graph_networkx(n)
        N = G.number_of_nodes()
        x = torch.from_numpy(rng.random((N, 1)).astype(np.float32))
        with torch.no_grad():
            y = ref_model(G, x)
        np.savez_compressed(os.path.join(GOLD, f"model_n{n}.npz"), x=x.numpy(), y=y.numpy(),
                            model_seed=1234, sd_seed=99, sd_checksum=np.float64(checksum),
                            n_params=sum(p.numel() for p in ref_model.parameters()))
    gen_train(ref)
    gen_progress(ref)
    print("golden vectors written to", GOLD)


TRAIN_CASES = ((5, 3), (8, 2))            # (n, batch)
SAMPLE = 129                              # gradient entries kept per parameter (stride sample) + sum / abs-sum


def grad_digest(g):
    """Compact pin of one gradient tensor: fp64 sum, abs-sum and a strided sample of its entries."""
    import numpy as np
    flat = g.detach().double().reshape(-1).numpy()
    idx = np.unique(np.linspace(0, flat.size - 1, min(SAMPLE, flat.size)).astype(np.int64))
    return np.float64(flat.sum()), np.float64(np.abs(flat).sum()), idx.astype(np.int32), flat[idx].astype(np.float32)


def gen_train(ref):
    """(9) one training step (train.py:20-32: model.train(), MSELoss, loss.backward()) through the reference's own
    models.py on a dgl.batch-style disjoint union of line graphs: predictions, loss, a digest of every parameter
    gradient and the BatchNorm running statistics after the step -> tests/golden/train_n{n}.npz."""
    import torch
    from oracle import model_oracle as mo

    for n, batch in TRAIN_CASES:
        torch.manual_seed(4321)
        model = ref.models.EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
        sd = mo.synthetic_state_dict(model, seed=77)
        model.load_state_dict(sd)
        rng = np.random.default_rng(1000 + n)
        N = n * (n - 1) // 2
        x = torch.from_numpy(rng.random((batch * N, 1)).astype(np.float32))
        target = torch.from_numpy(rng.random((batch * N, 1)).astype(np.float32))
        G = mo.batch_line_graphs(n, batch)
        y, loss, grads, bufs = mo.train_step_reference(model, G, x, target)
        out = dict(x=x.numpy(), target=target.numpy(), y=y.numpy(), loss=np.float32(loss.item()), n=n, batch=batch,
                   model_seed=4321, sd_seed=77)
        for k, g in grads.items():
            s, a, idx, val = grad_digest(g)
            out["gsum/" + k], out["gabs/" + k], out["gidx/" + k], out["gval/" + k] = s, a, idx, val
        for k, b in bufs.items():
            out["buf/" + k] = b.numpy()
        np.savez_compressed(os.path.join(GOLD, f"train_n{n}.npz"), **out)


PROGRESS_CASES = [
    # n, K, guides, perturbation_moves -- start tour as scripts/test.py:70-88 builds it: greedy on 'regret_pred' whenever
    # that guide is used at all (also when it is not the first one), else on 'weight'
    (20, 60, ["weight", "regret_pred"], 20),
    (50, 40, ["weight"], 20),
    (50, 25, ["weight", "regret_pred"], 10),
    (100, 30, ["regret_pred"], 20),
]


def gen_progress(ref):
    """(10) the trajectory of the RETURNED best (algorithms.py:143,190-191) under the fake clock: the cost every
    local_search call of the reference returns is captured, and the improvement record = its strict running minima
    (call 0 = the initial descent -> iteration 0, call k -> k completed outer iterations) plus a terminal entry
    -> tests/golden/gls_imp_c*.npz (pins the oracle's / the device's improvement trace)."""
    rng = np.random.default_rng(20261002)
    alg = ref.algorithms
    for ci, (n, K, guides, pm) in enumerate(PROGRESS_CASES):
        G = make_graph(rng.random((n, 2)))
        for e in G.edges:
            G.edges[e]["regret_pred"] = np.maximum(np.float32(rng.normal(0.05, 0.1)).item(), 0)
        D = np.asarray(nx.attr_matrix(G, "weight")[0])
        W = {g: np.asarray(nx.attr_matrix(G, g)[0]) for g in guides}
        init_guide = "regret_pred" if "regret_pred" in guides else "weight"              # test.py:70,85,88
        init_tour = alg.nearest_neighbor(G, 0, weight=init_guide)
        init_cost = ref.tour_cost(G, init_tour)
        ls_costs = []
        real_ls = alg.local_search

        def recording_ls(*a, **kw):
            r = real_ls(*a, **kw)
            ls_costs.append(r[1])
            return r

        alg.local_search = recording_ls
        try:
            best_tour, best_cost, trace, pen = run_ref_gls(ref, G, list(init_tour), init_cost, K, guides, pm, False)
        finally:
            alg.local_search = real_ls
        assert len(ls_costs) == K + 1
        imp_cost, imp_iter, best = [], [], None
        for k, c in enumerate(ls_costs):
            if best is None or c < best:
                best = c
                imp_cost.append(c)
                imp_iter.append(k)
        assert best == best_cost
        imp_cost.append(best_cost)
        imp_iter.append(K)
        np.savez_compressed(
            os.path.join(GOLD, f"gls_imp_c{ci}_n{n}_K{K}.npz"),
            D=D, guides=np.stack([W[g] for g in guides]), guide_names=np.array(guides), init_guide=np.array(init_guide),
            nn_guide=np.asarray(nx.attr_matrix(G, init_guide)[0]),
            init_tour=np.array(init_tour, dtype=np.int32), init_cost=np.float64(init_cost),
            K=K, perturbation_moves=pm, first_improvement=0,
            best_tour=np.array(best_tour, dtype=np.int32), best_cost=np.float64(best_cost),
            trace=np.array(trace, dtype=np.float64), penalty=pen.astype(np.int32),
            imp_cost=np.array(imp_cost, dtype=np.float64), imp_iter=np.array(imp_iter, dtype=np.int64))
        print(f"gls_imp_c{ci}: n={n} K={K} moves={len(trace)} improvements={len(imp_cost) - 1}")


if __name__ == "__main__":
    if "--only-progress" in sys.argv:       # adds the improvement-record fixtures without rewriting the others
        gen_progress(ref_import.import_reference())
    elif "--only-train" in sys.argv:          # adds the training fixtures without rewriting the others
        gen_train(ref_import.import_reference(with_models=True))
    else:
        main()
