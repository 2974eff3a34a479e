"""SURVEY 8(f) N3 on the CPU: the reference's on-disk formats (networkx-2.8 instance pickles, scikit-learn-1.0.2 scalers in
both layouts of datasets.py:48-51, a DGL-0.6.1-shaped checkpoint as scripts/train.py:59-66 writes it) go through the loaders
of gnngls_amd under this image's networkx 3.x / scikit-learn 1.7.  Fixture: tests/golden/n3_tsp12/ (made by
tests/golden/make_n3_fixtures.py; data only).  The end-to-end run through scripts/test.py is tests/test_n3_ingestion_gpu.py."""
import os
import pickle
import pickletools
import warnings

import networkx as nx
import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "n3_tsp12")


def names():
    return open(os.path.join(FIX, "test.txt")).read().split()


def test_fixture_really_has_the_networkx_28_layout():
    raw = open(os.path.join(FIX, names()[0]), "rb").read()
    ops = [(op.name, arg) for op, arg, _ in pickletools.genops(raw)]
    strings = {arg for name, arg in ops if isinstance(arg, str)}
    assert ops[0] == ("PROTO", 5)                                            # python 3.8's HIGHEST_PROTOCOL (nx.write_gpickle)
    assert "__networkx_cache__" not in strings                               # that slot is networkx 3.x
    assert {"_adj", "_node", "graph", "nodes", "edges", "adj"} <= strings     # cached views sit in the instance dict
    assert "networkx.classes.graph" in strings and "networkx.classes.reportviews" in strings
    G = pickle.load(open(os.path.join(FIX, names()[0]), "rb"))               # plain unpickling: what nx 3.x makes of it
    assert "__networkx_cache__" not in G.__dict__ and "edges" in G.__dict__


def test_loader_completes_the_graph_and_everything_downstream_works():
    from gnngls_amd import datasets, host
    from oracle import held_karp
    G = datasets.read_gpickle(os.path.join(FIX, names()[0]))
    assert G.__dict__["__networkx_cache__"] == {} and len(G.nodes) == 12 and len(G.edges) == 66
    e = next(iter(G.edges))
    assert isinstance(G.edges[e]["weight"], np.float64) and G.edges[e]["features"].dtype == np.float32
    assert isinstance(G.nodes[0]["pos"], np.ndarray)
    # what scripts/test.py does with an instance: optimum from the labels, attribute writes, matrices, mutation
    D = np.asarray(nx.attr_matrix(G, "weight", rc_order=sorted(G.nodes)))
    opt, tour = held_karp.optimum(D)
    assert host.optimal_cost(G) == pytest.approx(opt, rel=1e-12) and host.is_valid_tour(G, tour)
    G.edges[e]["regret_pred"] = 0.5
    nx.set_edge_attributes(G, 0, "penalty")
    G.add_edge(0, 1, weight=G.edges[0, 1]["weight"])
    assert len(nx.line_graph(G)) == 66 and G.copy().number_of_edges() == 66


@pytest.mark.parametrize("scalers_file", [None, "scalers_edges_layout.pkl"])
def test_dataset_reads_both_scaler_layouts_written_by_sklearn_102(scalers_file):
    from gnngls_amd import datasets
    from gnngls_amd.pipeline import Scalers
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                                      # InconsistentVersionWarning: 1.0.2 pickle under 1.7
        ds = datasets.TSPDataset(os.path.join(FIX, "test.txt"), None if scalers_file is None else os.path.join(FIX, scalers_file))
    assert set(ds.scalers) == {"features", "regret"} and len(ds) == 3 and ds.G.n == 12
    sc = Scalers.from_sklearn(ds.scalers)
    H = ds[1]
    G = datasets.read_gpickle(os.path.join(FIX, names()[1]))
    es = ds.G.ndata["e"].numpy()
    w32 = np.array([G.edges[tuple(e)]["features"][0] for e in es], dtype=np.float32)
    # sklearn's transform on fp32 input: x * scale_ + min_ evaluated in fp32 (what gnngls_pack_features reproduces)
    want = (w32 * np.float32(ds.scalers["features"].scale_[0]) + np.float32(ds.scalers["features"].min_[0])).astype(np.float32)
    got = H.ndata["features"][:, 0].numpy()
    assert np.abs(got - want).max() <= 1e-6 and 0.0 <= got.min() and got.max() <= 1.0 + 1e-6
    assert sc.feat_scale == float(ds.scalers["features"].scale_[0]) and sc.regret_min == float(ds.scalers["regret"].min_[0])
    assert H.ndata["regret"].shape == (66, 1)


def dgl061_checkpoint(seed=0):
    """A checkpoint dict exactly as scripts/train.py:59-66 saves it from the reference model under dgl 0.6.1: GATConv has NO
    bias (it appeared in 0.7), BatchNorm carries num_batches_tracked, Adam's state sits beside the weights."""
    g = torch.Generator().manual_seed(seed)
    sd = {"embed_layer.weight": torch.randn(128, 1, generator=g) * 0.5, "embed_layer.bias": torch.randn(128, generator=g) * 0.1}
    for k in range(8):
        p = f"message_passing_layers.{k}."
        sd[p + "message_passing.module.fc.weight"] = torch.randn(128, 128, generator=g) * 0.12
        sd[p + "message_passing.module.attn_l"] = torch.randn(1, 8, 16, generator=g) * 0.3
        sd[p + "message_passing.module.attn_r"] = torch.randn(1, 8, 16, generator=g) * 0.3
        for bn in ("feed_forward.0.", "feed_forward.2."):
            sd[p + bn + "weight"] = 1.0 + 0.1 * torch.randn(128, generator=g)
            sd[p + bn + "bias"] = 0.1 * torch.randn(128, generator=g)
            sd[p + bn + "running_mean"] = 0.1 * torch.randn(128, generator=g)
            sd[p + bn + "running_var"] = 0.5 + torch.rand(128, generator=g)
            sd[p + bn + "num_batches_tracked"] = torch.tensor(1234, dtype=torch.long)
        sd[p + "feed_forward.1.module.0.weight"] = torch.randn(512, 128, generator=g) * 0.08
        sd[p + "feed_forward.1.module.0.bias"] = torch.randn(512, generator=g) * 0.05
        sd[p + "feed_forward.1.module.2.weight"] = torch.randn(128, 512, generator=g) * 0.04
        sd[p + "feed_forward.1.module.2.bias"] = torch.randn(128, generator=g) * 0.05
    sd["decision_layer.weight"] = torch.randn(1, 128, generator=g) * 0.1
    sd["decision_layer.bias"] = torch.randn(1, generator=g) * 0.1
    params = [v for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k]
    opt = {"state": {i: {"step": torch.tensor(10.0), "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
                     for i, p in enumerate(params)},
           "param_groups": [{"lr": 1e-3, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                             "params": list(range(len(params)))}]}
    return {"epoch": 37, "model_state_dict": sd, "optimizer_state_dict": opt, "loss": 0.0123, "val_loss": 0.0131}


def test_dgl_061_shaped_checkpoint_loads_strictly():
    """The key set SURVEY 8(a) a1 lists (1,191,297 parameters, no GATConv bias) is exactly the mirror model's."""
    from gnngls_amd.models import EdgePropertyPredictionModel
    ck = dgl061_checkpoint()
    model = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
    assert set(model.state_dict()) == set(ck["model_state_dict"])
    model.load_state_dict(ck["model_state_dict"])                            # strict
    assert sum(p.numel() for p in model.parameters()) == 1191297 and model.gat_biases() == []
    assert int(model.message_passing_layers[3].feed_forward[0].num_batches_tracked) == 1234


def test_lfs_pointer_stubs_are_reported(tmp_path):
    from gnngls_amd import datasets
    p = tmp_path / "checkpoint_best_val.pt"
    p.write_text("version https://git-lfs.github.com/spec/v1\noid sha256:abc\nsize 14300000\n")
    assert datasets.is_lfs_pointer(p)
    with pytest.raises(FileNotFoundError, match="git-LFS pointer"):
        datasets.read_gpickle(p)
    with pytest.raises(FileNotFoundError, match="git-LFS pointer"):
        datasets.TSPDataset(p)
