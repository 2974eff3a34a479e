"""CPU tests of the host-side mirror: pure helpers, list transforms, weight packing, dataset surface."""
import itertools
import os
import pickle

import networkx as nx
import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_two_opt_relocate_list_transforms_match_golden():
    from gnngls_amd import operators as ops
    g = np.load(os.path.join(GOLD, "ops_n20.npz"))
    tour = g["tour"].tolist()
    n = len(tour) - 1
    from oracle import gls_oracle as go
    for i, j in itertools.product(range(1, n), repeat=2):
        assert ops.two_opt(tour, i, j) == go.two_opt(tour, i, j)
        assert ops.relocate(tour, i, j) == go.relocate(tour, i, j)
    assert ops.two_opt(tour, 3, 3) is tour


def make_graph(n, rng):
    G = nx.Graph()
    pos = rng.random((n, 2))
    for k, p in enumerate(pos):
        G.add_node(k, pos=p)
    for i, j in itertools.combinations(G.nodes, 2):
        G.add_edge(i, j, weight=np.linalg.norm(pos[j] - pos[i]))
    return G


def test_host_helpers_match_golden_and_reference_semantics():
    import gnngls_amd
    rng = np.random.default_rng(0)
    G = make_graph(7, rng)
    tour = [0, 3, 1, 2, 6, 5, 4, 0]
    c = gnngls_amd.tour_cost(G, tour)
    assert c == sum(G.edges[e]["weight"] for e in zip(tour[:-1], tour[1:]))
    assert gnngls_amd.is_valid_tour(G, tour)
    assert not gnngls_amd.is_valid_tour(G, tour[:-1] + [1])
    assert not gnngls_amd.is_valid_tour(G, [0, 1, 1, 2, 3, 4, 5, 0])
    assert gnngls_amd.is_equivalent_tour(tour, tour[::-1])
    nx.set_edge_attributes(G, gnngls_amd.tour_to_edge_attribute(G, tour), "in_solution")
    assert gnngls_amd.optimal_cost(G) == pytest.approx(c, rel=1e-12)
    g = np.load(os.path.join(GOLD, "misc.npz"))
    from gnngls_amd.algorithms import _attr_matrix
    n = g["nn_W_weight"].shape[0]
    H = nx.Graph()
    H.add_nodes_from(range(n))
    for i, j in itertools.combinations(range(n), 2):
        H.add_edge(i, j, weight=np.float64(g["nn_W_weight"][i, j]), in_solution=bool(g["opt_in_solution"][i, j]))
    assert np.array_equal(_attr_matrix(H, "weight"), g["nn_W_weight"])
    assert np.float64(gnngls_amd.tour_cost(H, g["tc_tour"].tolist())).tobytes() == g["tc_cost"].tobytes()
    assert np.float64(gnngls_amd.optimal_cost(H)).tobytes() == g["opt_cost"].tobytes()


def test_line_graph_and_state_dict_layout():
    from gnngls_amd.models import EdgePropertyPredictionModel, LineGraph
    from oracle import model_oracle as mo
    G = LineGraph(6, batch=3)
    assert G.number_of_nodes() == 45 and G.ndata["e"].shape == (45, 2)
    assert G.ndata["e"][:15].tolist() == [list(e) for e in itertools.combinations(range(6), 2)]
    torch.manual_seed(0)
    model = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
    oracle = mo.EdgeRegretModelOracle(1, 128, 1, 3, n_heads=8)
    assert len(model.message_passing_layers) == 8              # models.py:59-61
    sd_o = oracle.state_dict()
    assert list(model.state_dict().keys()) == list(sd_o.keys())
    assert all(model.state_dict()[k].shape == v.shape for k, v in sd_o.items())
    assert sum(p.numel() for p in model.parameters()) == 1191297
    model.load_state_dict(sd_o)
    packed = model.eval().pack_weights("cpu")
    assert packed.dtype == torch.float32 and packed.numel() % 4 == 0
    # spot-check the documented layout (include/gnngls_hip.h)
    assert torch.equal(packed[:128], sd_o["embed_layer.weight"].reshape(-1))
    assert torch.equal(packed[128:256], sd_o["embed_layer.bias"])
    assert torch.equal(packed[256:256 + 128 * 128], sd_o["message_passing_layers.0.message_passing.module.fc.weight"].reshape(-1))
    assert torch.equal(packed[-132:-4], sd_o["decision_layer.weight"].reshape(-1))
    with pytest.raises(NotImplementedError):
        EdgePropertyPredictionModel(1, 64, 1, 3, n_heads=4).pack_weights("cpu")
    from gnngls_amd import _lib
    for mode in (model.train(), model.eval()):          # no CPU path in either mode
        with pytest.raises(_lib.GnnglsHipError):
            mode(G, torch.zeros(45, 1))
    # order of the raw training image (include/gnngls_hip.h, N4) = order of the packed inference image
    tp = model.train_parameters()
    assert sum(p.numel() for p in tp) + 3 == packed.numel() and tp[2] is model.message_passing_layers[0].message_passing.module.fc.weight


def test_dataset_surface(tmp_path):
    from sklearn.preprocessing import MinMaxScaler

    from gnngls_amd import datasets
    from gnngls_amd.pipeline import Scalers
    rng = np.random.default_rng(1)
    names = []
    scalers = {"features": MinMaxScaler(), "regret": MinMaxScaler()}
    for k in range(3):
        G = make_graph(6, rng)
        datasets.set_features(G)
        for e in G.edges:
            G.edges[e]["regret"] = float(rng.random())
            G.edges[e]["in_solution"] = False
        for key in scalers:
            scalers[key].partial_fit(np.vstack([G.edges[e][key] for e in G.edges]))
        name = f"inst{k}.pkl"
        pickle.dump(G, open(tmp_path / name, "wb"))
        names.append(name)
    (tmp_path / "test.txt").write_text("\n".join(names) + "\n")
    pickle.dump({"edges": scalers}, open(tmp_path / "scalers.pkl", "wb"))
    ds = datasets.TSPDataset(tmp_path / "test.txt")
    assert len(ds) == 3 and ds.G.n == 6 and ds.root_dir == tmp_path
    H = ds[1]
    G1 = datasets.read_gpickle(tmp_path / names[1])
    es = [tuple(e) for e in ds.G.ndata["e"].tolist()]
    expect = scalers["features"].transform(np.vstack([G1.edges[e]["features"] for e in es]))
    assert H.ndata["features"].dtype == torch.float32 and np.array_equal(H.ndata["features"].numpy(), expect)
    assert H.ndata["regret"].shape == (15, 1) and H.ndata["in_solution"].shape == (15, 1)
    s = Scalers.from_sklearn({"edges": scalers})
    assert s.feat_scale == scalers["features"].scale_[0] and s.regret_min == scalers["regret"].min_[0]
    # git-LFS pointer stubs (every data/ and models/ file of the reference) are reported, not unpickled
    (tmp_path / "stub.pkl").write_bytes(b"version https://git-lfs.github.com/spec/v1\noid sha256:00\nsize 1\n")
    assert datasets.is_lfs_pointer(tmp_path / "stub.pkl")
    with pytest.raises(FileNotFoundError):
        datasets.read_gpickle(tmp_path / "stub.pkl")


def test_synthetic_instances_are_symmetric_euclidean():
    from gnngls_amd.synthetic import random_instances
    D, pos = random_instances(np.random.default_rng(3), 4, 9)
    assert D.shape == (4, 9, 9) and np.array_equal(D, D.transpose(0, 2, 1)) and (np.diagonal(D, axis1=1, axis2=2) == 0).all()
    assert np.allclose(D[2, 1, 5], np.linalg.norm(pos[2, 1] - pos[2, 5]), rtol=1e-15)


def test_cli_progress_rows_always_reach_the_returned_cost():
    """scripts/test.py progress_rows: whatever the per-move record holds (nothing at all when no move was accepted; a
    complete record whose minimum is above a never-improved start cost; a truncated one), the rows end on the terminal
    entry of the improvement record, so the cummin `best_cost` column reaches the returned cost."""
    import importlib.util
    import os
    import types
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gnngls_cli_rows", os.path.join(root, "scripts", "test.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)

    def res(moves, trace, imp):
        r = types.SimpleNamespace()
        r.moves = torch.tensor([moves])
        r.trace_time = torch.tensor([[0.1 * (i + 1) for i in range(len(trace))] + [0.0]], dtype=torch.float32)
        r.trace_cost = torch.tensor([list(trace) + [0.0]], dtype=torch.float64)
        r.imp_time = torch.tensor([[t for t, _ in imp]], dtype=torch.float32)
        r.imp_cost = torch.tensor([[c for _, c in imp]], dtype=torch.float64)
        r.imp_len = torch.tensor([len(imp)])
        return r

    # no accepted move at all: start tour already locally optimal, budget over after the first descent
    rows, cut = cli.progress_rows(res(0, [], [(0.0, 7.5), (0.01, 7.5)]), 0, full_trace=100)
    assert not cut and rows and min(c for _, c in rows) == 7.5
    # complete record whose costs all lie above the returned best (the best is the cost after the initial descent = start)
    rows, cut = cli.progress_rows(res(3, [8.0, 7.9, 7.7], [(0.0, 7.5), (1.0, 7.5)]), 0, full_trace=100)
    assert not cut and len(rows) == 4 and rows[-1][1] == 7.5 and abs(rows[-1][0] - 1.0) < 1e-6
    # truncated record: improvement events after the cut, then the terminal entry
    rows, cut = cli.progress_rows(res(9, [8.0, 7.9], [(0.05, 7.9), (0.5, 7.2), (0.9, 7.0), (1.0, 7.0)]), 0, full_trace=2)
    assert cut and [c for _, c in rows] == [8.0, 7.9, 7.2, 7.0, 7.0]
    # default record (no per-move trace): improvement events + terminal
    rows, cut = cli.progress_rows(res(9, [], [(0.05, 7.9), (0.5, 7.2), (1.0, 7.2)]), 0, full_trace=0)
    assert not cut and [c for _, c in rows] == [7.9, 7.2, 7.2]
