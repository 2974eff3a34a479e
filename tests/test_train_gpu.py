"""GPU parity tests of one training step (scripts/train.py:20-32: model.train(), y_pred = model(batch, x),
loss.backward()) of the HIP path, through the C ABI, against (a) the fixtures captured from the reference's own models.py
(tests/golden/train_n*.npz, oracle/gen_golden.py) and (b) the CPU oracle differentiated by torch autograd on seeded
batches.

Tolerances.  Predictions: 1e-5 relative with a 1e-5*max|y| floor (BASELINE.json north_star).  Gradients are
ill-conditioned sums over all B*N rows, and the network has kinks (ReLU, LeakyReLU): a pre-activation within rounding
distance of 0 takes a different branch in different fp32 evaluations and moves the affected gradient rows by ~1e-3..1e-2 of
the tensor's largest entry.  A plain fp32 CPU evaluation of the reference graph (the fp32 oracle) shows exactly this against
the fp64 oracle, and over random cases its errors and the HIP path's are the same distribution (scripts/
train_parity_campaign.py prints both: whole-gradient relative L2 error 4e-6 .. 1e-3 for either).  The bar is therefore
"indistinguishable from the fp32 evaluation of the reference graph", per case (gradient_errors_acceptable):
    whole-gradient relative L2 error      <= max(5e-4, 3 x fp32 oracle's)
    worst per-tensor relative L2 error    <= max(3e-3, 3 x fp32 oracle's)
    worst per-tensor max error / max|g|   <= max(2e-2, 3 x fp32 oracle's)
    median over tensors of that           <= max(5e-4, 3 x fp32 oracle's)
    exact-zero gradients (a Linear bias feeding a BatchNorm) stay below max(1e-6, 3 x fp32 oracle's) of the largest entry
and over a set of cases the medians of the HIP errors must not exceed 1.5 x the fp32 oracle's (test_gradient_error_statistics).
"""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_models(seed, sd_seed):
    from gnngls_amd.models import EdgePropertyPredictionModel
    from oracle import model_oracle as mo
    torch.manual_seed(seed)
    oracle = mo.EdgeRegretModelOracle(1, 128, 1, 3, n_heads=8)
    sd = mo.synthetic_state_dict(oracle, seed=sd_seed)
    oracle.load_state_dict(sd)
    model = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
    model.load_state_dict(sd)
    return model.to("cuda"), oracle


def hip_step(model, n, B, x, target, criterion=None):
    from gnngls_amd.models import LineGraph
    criterion = criterion or torch.nn.MSELoss()
    model.train()
    model.zero_grad()
    y = model(LineGraph(n, batch=B).to("cuda"), x.cuda())
    loss = criterion(y, target.cuda().type_as(y))
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    bufs = {k: b.detach().cpu() for k, b in model.named_buffers()}
    return y.detach().cpu(), loss.item(), grads, bufs


def gradient_error_metrics(g_hip, g32, g64):
    """Errors of the HIP gradients and of the fp32 CPU oracle's against the fp64 oracle: relative L2 over the whole
    parameter vector, and per tensor the relative L2 and max errors (tensors whose exact gradient is zero -- a Linear bias
    feeding a BatchNorm -- are left out of the per-tensor statistics)."""
    def l2(a, b):
        return float((a.double() - b).pow(2).sum())
    tot = sum(float(v.pow(2).sum()) for v in g64.values())
    gmax = max(v.abs().max().item() for v in g64.values())
    live = [k for k, v in g64.items() if v.abs().max().item() > 1e-9 * gmax]
    out = {"global_l2_hip": (sum(l2(g_hip[k], v) for k, v in g64.items()) / tot) ** 0.5,
           "global_l2_32": (sum(l2(g32[k], v) for k, v in g64.items()) / tot) ** 0.5}
    for name, g in (("hip", g_hip), ("32", g32)):
        tl2 = [(l2(g[k], g64[k]) / float(g64[k].pow(2).sum())) ** 0.5 for k in live]
        tmax = [(g[k].double() - g64[k]).abs().max().item() / g64[k].abs().max().item() for k in live]
        out["worst_l2_" + name], out["worst_max_" + name], out["median_max_" + name] = max(tl2), max(tmax), float(np.median(tmax))
    dead = [k for k in g64 if k not in live]
    out["dead_abs_hip"] = max([g_hip[k].abs().max().item() for k in dead], default=0.0) / gmax
    out["dead_abs_32"] = max([g32[k].abs().max().item() for k in dead], default=0.0) / gmax
    return out


def gradient_errors_acceptable(m):
    """The per-case bounds of the module docstring."""
    return (m["global_l2_hip"] <= max(5e-4, 3 * m["global_l2_32"])
            and m["worst_l2_hip"] <= max(3e-3, 3 * m["worst_l2_32"])
            and m["worst_max_hip"] <= max(2e-2, 3 * m["worst_max_32"])
            and m["median_max_hip"] <= max(5e-4, 3 * m["median_max_32"])
            and m["dead_abs_hip"] <= max(1e-6, 3 * m["dead_abs_32"]))


def assert_pred_close(y, ref, rtol=1e-5):
    y, ref = np.asarray(y, np.float64).reshape(-1), np.asarray(ref, np.float64).reshape(-1)
    bound = rtol * np.abs(ref) + rtol * np.abs(ref).max()
    err = np.abs(y - ref)
    assert (err <= bound).all(), f"max err {err.max():.3e}, worst ratio {(err / bound).max():.2f}"


@pytest.mark.parametrize("n", [5, 8])
def test_train_step_golden(n):
    g = np.load(os.path.join(GOLD, f"train_n{n}.npz"))
    B = int(g["batch"])
    model, _ = make_models(int(g["model_seed"]), int(g["sd_seed"]))
    y, loss, grads, bufs = hip_step(model, n, B, torch.from_numpy(g["x"]), torch.from_numpy(g["target"]))
    # the fixture is the reference's own fp32 run on a tiny batch (BatchNorm statistics over 30 / 56 rows): coarse pin
    assert_pred_close(y.numpy(), g["y"], rtol=1e-4)
    assert abs(loss - float(g["loss"])) <= 1e-4 * float(g["loss"])
    for k, gr in grads.items():
        flat = gr.double().reshape(-1).numpy()
        ref = g["gval/" + k].astype(np.float64)
        # coarse pin against the reference run (itself fp32, see the module docstring); the tight bound is below
        tol = 5e-3 * np.abs(ref).max() + 1e-6
        assert np.abs(flat[g["gidx/" + k]] - ref).max() <= tol, k
    for k, b in bufs.items():
        if b.dtype.is_floating_point:
            assert np.allclose(b.numpy(), g["buf/" + k], rtol=2e-5, atol=1e-6), k
        else:
            assert np.array_equal(b.numpy(), g["buf/" + k]), k          # num_batches_tracked


@pytest.mark.parametrize("n,B", [(3, 5), (4, 3), (6, 2), (17, 2), (33, 2), (50, 1), (64, 2), (100, 1), (120, 1)])
def test_train_step_vs_oracle(n, B):
    from oracle import model_oracle as mo
    model, oracle = make_models(4321, 77)
    oracle64 = copy.deepcopy(oracle).double()
    N = n * (n - 1) // 2
    rng = np.random.default_rng(100 * n + B)
    x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    target = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    G = mo.batch_line_graphs(n, B)
    y32, loss32, g32, b32 = mo.train_step_reference(oracle, G, x, target)
    y64, loss64, g64, b64 = mo.train_step_reference(oracle64, G, x.double(), target.double())
    y, loss, grads, bufs = hip_step(model, n, B, x, target)

    err = np.abs(y.double().numpy() - y64.numpy()).reshape(-1)
    ref = np.abs(y64.numpy()).reshape(-1)
    own = np.abs(y32.double().numpy() - y64.numpy()).reshape(-1).max()
    assert (err <= 1e-5 * ref + 1e-5 * ref.max() + 3 * own).all(), f"pred err {err.max():.3e} (fp32 oracle {own:.3e})"
    assert abs(loss - loss64.item()) <= 1e-5 * loss64.item() + 3 * abs(loss32.item() - loss64.item())

    m = gradient_error_metrics(grads, g32, g64)
    assert gradient_errors_acceptable(m), m
    # running statistics after the step (models.py:27,35: momentum 0.1, unbiased batch variance)
    for k, ref64 in b64.items():
        if ref64.dtype.is_floating_point:
            assert torch.allclose(bufs[k].double(), ref64, rtol=2e-5, atol=1e-6), k
        else:
            assert torch.equal(bufs[k], ref64), k


def one_layer(m):
    """Keep the first AttentionLayer only (the C ABI takes the layer count; the oracle iterates what is there): the fp64
    autograd oracle of a TSP150 / TSP200 step stays within a few GB and a minute."""
    m.message_passing_layers = m.message_passing_layers[:1]
    return m


@pytest.mark.parametrize("n", [150, 200])
def test_train_step_vs_oracle_beyond_the_old_tile_limit(n):
    """n > 145 (TSP200 = BASELINE configs[4]): the attention backward's instantiations with 13 source tiles per row;
    forward through the head-split gat_rows_kernel.  One layer, one instance, every gradient tensor against the fp64
    oracle (arcs from the closed-form rule, checkpointed destination ranges: tests/test_model_oracle.py)."""
    from oracle import model_oracle as mo
    model, oracle = make_models(4321, 77)
    model, oracle = one_layer(model), one_layer(oracle)
    oracle64 = copy.deepcopy(oracle).double()
    N = n * (n - 1) // 2
    rng = np.random.default_rng(n)
    x = torch.from_numpy(rng.random((N, 1)).astype(np.float32))
    target = torch.from_numpy(rng.random((N, 1)).astype(np.float32))
    G = mo.line_graph_arcs_closed_form(n)
    y32, loss32, g32, b32 = mo.train_step_reference(oracle, G, x, target)
    y64, loss64, g64, b64 = mo.train_step_reference(oracle64, G, x.double(), target.double())
    y, loss, grads, bufs = hip_step(model, n, 1, x, target)
    own = np.abs(y32.double().numpy() - y64.numpy()).max()
    err, ref = np.abs(y.double().numpy() - y64.numpy()).reshape(-1), np.abs(y64.numpy()).reshape(-1)
    assert (err <= 1e-5 * ref + 1e-5 * ref.max() + 3 * own).all(), f"pred err {err.max():.3e} (fp32 oracle {own:.3e})"
    assert abs(loss - loss64.item()) <= 1e-5 * loss64.item() + 3 * abs(loss32.item() - loss64.item())
    m = gradient_error_metrics(grads, g32, g64)
    assert gradient_errors_acceptable(m), m
    for k, ref64 in b64.items():
        if ref64.dtype.is_floating_point:
            assert torch.allclose(bufs[k].double(), ref64, rtol=2e-5, atol=1e-6), k


def test_train_step_with_gatconv_bias():
    """A checkpoint with GATConv `bias` keys (DGL >= 0.7): training works, predictions and all other gradients equal the
    oracle's (whose GATConv adds the bias after the aggregation), the bias gradient is zero (BatchNorm-1 normalises with the
    batch mean) and the BatchNorm-1 running mean moves by the bias."""
    from gnngls_amd.models import EdgePropertyPredictionModel
    from oracle import model_oracle as mo
    torch.manual_seed(99)
    oracle = mo.EdgeRegretModelOracle(1, 128, 1, 3, n_heads=8, gat_bias=True)
    sd = mo.synthetic_state_dict(oracle, seed=5)
    gen = torch.Generator().manual_seed(6)
    for k in list(sd):
        if k.endswith("message_passing.module.bias"):
            sd[k] = 0.3 * torch.randn(sd[k].shape, generator=gen)
    oracle.load_state_dict(sd)
    model = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
    res = model.load_state_dict(sd)
    assert not res.missing_keys and not res.unexpected_keys
    model.to("cuda")
    assert len(model.gat_biases()) == 8 and all(b.is_cuda and b.requires_grad for _, b in model.gat_biases())
    n, B = 12, 3
    N = n * (n - 1) // 2
    rng = np.random.default_rng(12)
    x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    target = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    G = mo.batch_line_graphs(n, B)
    oracle64 = copy.deepcopy(oracle).double()
    y32, loss32, g32, b32 = mo.train_step_reference(oracle, G, x, target)
    y64, loss64, g64, b64 = mo.train_step_reference(oracle64, G, x.double(), target.double())
    y, loss, grads, bufs = hip_step(model, n, B, x, target)
    assert_pred_close(y.numpy(), y64.numpy(), rtol=3e-5)
    bias_keys = [k for k in g64 if k.endswith("message_passing.module.bias")]
    assert len(bias_keys) == 8
    gmax = max(v.abs().max().item() for v in g64.values())
    for k in bias_keys:
        assert grads[k].abs().max().item() == 0.0 and g64[k].abs().max().item() <= 1e-9 * gmax        # exactly zero / rounding
    assert gradient_errors_acceptable(gradient_error_metrics(grads, g32, g64))
    for k, ref64 in b64.items():
        if ref64.dtype.is_floating_point:
            assert torch.allclose(bufs[k].double(), ref64, rtol=2e-5, atol=1e-6), k          # incl. running_mean shifted by the bias
    # and the same checkpoint in inference mode (bias folded into BatchNorm-1's shift)
    from gnngls_amd.models import LineGraph
    with torch.no_grad():
        ye = model.eval()(LineGraph(n, batch=B).to("cuda"), x.cuda()).cpu().numpy()
        re = oracle64.eval()(G, x.double()).numpy()
    assert_pred_close(ye, re)


def test_gradient_error_statistics():
    """Over a set of seeded random cases the HIP gradient errors are not larger than the fp32 CPU oracle's (medians)."""
    from oracle import model_oracle as mo
    rng = np.random.default_rng(2718)
    hip, ref = [], []
    for _ in range(10):
        n, B = int(rng.integers(3, 26)), int(rng.integers(1, 4))
        model, oracle = make_models(4321, 77)
        oracle64 = copy.deepcopy(oracle).double()
        N = n * (n - 1) // 2
        x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
        t = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
        G = mo.batch_line_graphs(n, B)
        _, _, g32, _ = mo.train_step_reference(oracle, G, x, t)
        _, _, g64, _ = mo.train_step_reference(oracle64, G, x.double(), t.double())
        _, _, gh, _ = hip_step(model, n, B, x, t)
        m = gradient_error_metrics(gh, g32, g64)
        assert gradient_errors_acceptable(m), (n, B, m)
        hip.append((m["global_l2_hip"], m["median_max_hip"]))
        ref.append((m["global_l2_32"], m["median_max_32"]))
    hip, ref = np.array(hip), np.array(ref)
    assert (np.median(hip, axis=0) <= 1.5 * np.median(ref, axis=0) + 1e-6).all(), (np.median(hip, axis=0), np.median(ref, axis=0))


def test_forward_workspace_is_owned_by_the_autograd_node():
    """Two training forwards before the first backward: the first node's activations must not be overwritten."""
    from gnngls_amd.models import LineGraph
    n, B = 9, 2
    N = n * (n - 1) // 2
    model, _ = make_models(4321, 77)
    rng = np.random.default_rng(7)
    x1 = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32)).cuda()
    t = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32)).cuda()
    G = LineGraph(n, batch=B).to("cuda")
    model.train()
    model.zero_grad()
    torch.nn.functional.mse_loss(model(G, x1), t).backward()
    single = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    y1 = model(G, x1)
    _ = model(G, x2)
    torch.nn.functional.mse_loss(y1, t).backward()
    for a, b in zip(single, [p.grad for p in model.parameters()]):
        assert torch.equal(a, b)                 # the backward is deterministic (fixed-order reductions)
    # a second backward through the same node has nothing to differentiate with: loud error, not stale numbers
    loss = torch.nn.functional.mse_loss(model(G, x1), t)
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="consumed"):
        loss.backward()


def test_adam_steps_track_the_oracle():
    """Three Adam steps (train.py:104,31) on the HIP path and on the CPU oracle: losses stay together, and the eval-mode
    forward picks up the updated parameters and running statistics (the packed inference image is rebuilt)."""
    from gnngls_amd.models import LineGraph
    from oracle import model_oracle as mo
    n, B = 10, 3
    N = n * (n - 1) // 2
    model, oracle = make_models(4321, 77)
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    target = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    G_cpu, G = mo.batch_line_graphs(n, B), LineGraph(n, batch=B).to("cuda")
    opt_h = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt_o = torch.optim.Adam(oracle.parameters(), lr=1e-3)
    crit = torch.nn.MSELoss()
    model.train(); oracle.train()
    for step in range(3):
        opt_h.zero_grad(); opt_o.zero_grad()
        lh = crit(model(G, x.cuda()), target.cuda()); lh.backward(); opt_h.step()
        lo = crit(oracle(G_cpu, x), target); lo.backward(); opt_o.step()
        # Adam normalises every gradient entry by its own magnitude: entries at rounding level move by +-lr in either
        # implementation, so the trajectories separate at the 1e-3 level per step -- track, do not match
        assert abs(lh.item() - lo.item()) <= 1e-2 * abs(lo.item()), (step, lh.item(), lo.item())
    # (eval-mode outputs are NOT compared with the oracle's: Adam turns the rounding-noise gradient of every Linear bias
    # that feeds a BatchNorm into +-lr steps, which training-mode BatchNorm cancels and eval-mode BatchNorm does not)
    from gnngls_amd.models import EdgePropertyPredictionModel
    with torch.no_grad():
        y_before = model.eval()(G, x.cuda())
        fresh = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
        fresh.load_state_dict(model.state_dict())
        y_fresh = fresh.to("cuda").eval()(G, x.cuda())
    assert torch.equal(y_before, y_fresh)                 # the cached inference image followed the optimizer
    model.train()
    crit(model(G, x.cuda()), target.cuda()).backward(); opt_h.step()
    with torch.no_grad():
        assert not torch.equal(model.eval()(G, x.cuda()), y_before)


def test_training_rejects_unsupported_sizes():
    from gnngls_amd import _lib
    from gnngls_amd.models import LineGraph
    model, _ = make_models(4321, 77)
    model.train()
    n = 258                                     # beyond the attention-backward tile limit (n <= 257)
    x = torch.zeros((n * (n - 1) // 2, 1), device="cuda")
    with pytest.raises(_lib.GnnglsHipError):
        model(LineGraph(n).to("cuda"), x)


def test_train_cli_end_to_end(tmp_path):
    """scripts/train.py with the reference's arguments on a tiny synthetic dataset: the loss goes down, the checkpoints
    and params.json appear with the reference's keys, and scripts/test.py's loader accepts the result."""
    import itertools
    import json
    import pickle
    import subprocess
    import sys

    import networkx as nx
    from sklearn.preprocessing import MinMaxScaler

    from gnngls_amd import datasets
    from gnngls_amd.models import EdgePropertyPredictionModel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(3)
    data = tmp_path / "tsp10"
    data.mkdir()
    scalers = {"features": MinMaxScaler(), "regret": MinMaxScaler()}
    names = []
    for k in range(12):
        pos = rng.random((10, 2))
        G = nx.Graph()
        for v, p in enumerate(pos):
            G.add_node(v, pos=p)
        for i, j in itertools.combinations(G.nodes, 2):
            w = np.linalg.norm(pos[j] - pos[i])
            G.add_edge(i, j, weight=w, in_solution=False, regret=float(w * w))     # a learnable target
        for v in range(10):
            G.edges[v, (v + 1) % 10]["in_solution"] = True
        datasets.set_features(G)
        for key in scalers:
            scalers[key].partial_fit(np.vstack([G.edges[e][key] for e in G.edges]))
        pickle.dump(G, open(data / f"i{k}.pkl", "wb"))
        names.append(f"i{k}.pkl")
    (data / "train.txt").write_text("\n".join(names[:8]) + "\n")
    (data / "val.txt").write_text("\n".join(names[8:]) + "\n")
    pickle.dump(scalers, open(data / "scalers.pkl", "wb"))
    tb = tmp_path / "tb"
    cmd = [sys.executable, os.path.join(root, "scripts", "train.py"), str(data), str(tb), "--batch_size", "4",
           "--n_epochs", "6", "--checkpoint_freq", "2", "--use_gpu", "--num_workers", "0"]
    subprocess.check_call(cmd, cwd=root)
    runs = list(tb.iterdir())
    assert len(runs) == 1
    files = {p.name for p in runs[0].iterdir()}
    assert {"checkpoint_best_val.pt", "checkpoint_final.pt", "checkpoint_2.pt", "checkpoint_4.pt", "params.json",
            "scalars.jsonl"} <= files
    params = json.load(open(runs[0] / "params.json"))
    assert params["embed_dim"] == 128 and params["n_heads"] == 8 and params["target"] == "regret"
    ck = torch.load(runs[0] / "checkpoint_final.pt", map_location="cpu")
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss", "val_loss"} and ck["epoch"] == 5
    model = EdgePropertyPredictionModel(1, params["embed_dim"], 1, params["n_layers"], n_heads=params["n_heads"])
    model.load_state_dict(ck["model_state_dict"])                                   # test.py:50-53
    scal = [json.loads(line) for line in open(runs[0] / "scalars.jsonl")]
    train_loss = [r["value"] for r in scal if r["tag"] == "Loss/train"]
    assert len(train_loss) == 6 and train_loss[-1] < 0.5 * train_loss[0]
    assert int(ck["model_state_dict"]["message_passing_layers.0.feed_forward.0.num_batches_tracked"]) == 6 * 2


def test_whole_step_hip_graph_replay_matches_eager():
    """torch.cuda.graphs whole-step capture (forward, loss, backward, Adam): every launch of the C ABI lands on torch's
    capturing stream, so the replayed steps must reproduce the eager steps bit for bit (the step is deterministic)."""
    from gnngls_amd.models import LineGraph
    n, B = 8, 2
    N = n * (n - 1) // 2
    rng = np.random.default_rng(21)
    x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32)).cuda()
    t = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32)).cuda()
    G = LineGraph(n, batch=B).to("cuda")

    def make():
        model, _ = make_models(4321, 77)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True)

        def step():
            opt.zero_grad(set_to_none=False)
            loss = torch.nn.functional.mse_loss(model(G, x), t)
            loss.backward()
            opt.step()
            return loss
        return model, step

    _, eager = make()
    eager_losses = [eager().item() for _ in range(5)]

    model, step = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = step()
    replay_losses = []
    for _ in range(2):
        graph.replay()
        replay_losses.append(static_loss.item())
    assert replay_losses == eager_losses[3:5]
    assert int(model.batch_norms()[0].num_batches_tracked) == 5


def test_two_input_features_and_bce_target():
    """in_dim = 2 (TSPDataset keeps every feature column that is not dropped, datasets.py:86) and the in_solution target
    with BCEWithLogitsLoss(pos_weight) (train.py:109-112): forward and every gradient against the fp64 oracle."""
    from gnngls_amd.models import EdgePropertyPredictionModel, LineGraph
    from oracle import model_oracle as mo
    n, B = 9, 3
    N = n * (n - 1) // 2
    torch.manual_seed(5)
    oracle = mo.EdgeRegretModelOracle(2, 128, 1, 3, n_heads=8)
    sd = mo.synthetic_state_dict(oracle, seed=6)
    oracle.load_state_dict(sd)
    oracle64 = copy.deepcopy(oracle).double()
    model = EdgePropertyPredictionModel(2, 128, 1, 3, n_heads=8)
    model.load_state_dict(sd)
    model = model.cuda()
    rng = np.random.default_rng(9)
    x = torch.from_numpy(rng.random((B * N, 2)).astype(np.float32))
    y = torch.from_numpy((rng.random((B * N, 1)) < 0.2).astype(np.float32))
    pos_weight = len(y) / y.sum() - 1
    G = mo.batch_line_graphs(n, B)
    _, loss64, g64, _ = mo.train_step_reference(oracle64, G, x.double(), y.double(),
                                                torch.nn.BCEWithLogitsLoss(pos_weight=pos_weight.double()))
    _, loss32, g32, _ = mo.train_step_reference(oracle, G, x, y, torch.nn.BCEWithLogitsLoss(pos_weight=pos_weight))
    _, loss, grads, _ = hip_step(model, n, B, x, y, torch.nn.BCEWithLogitsLoss(pos_weight=pos_weight.cuda()))
    assert abs(loss - loss64.item()) <= 1e-5 * loss64.item() + 3 * abs(loss32.item() - loss64.item())
    m = gradient_error_metrics(grads, g32, g64)
    assert gradient_errors_acceptable(m), m
    assert grads["embed_layer.weight"].shape == (128, 2)
    # eval-mode forward with two input features
    model.eval(); oracle64.eval()
    with torch.no_grad():
        yh = model(LineGraph(n, batch=B).to("cuda"), x.cuda()).cpu().double()
        yo = oracle64(G, x.double())
    assert_pred_close(yh.numpy(), yo.numpy(), rtol=2e-5)


@pytest.mark.parametrize("scale", [10.0, 400.0])
def test_train_step_with_saturated_attention(scale):
    """Large attention logits: the factorised weights of gat_rows_kernel must stay finite and accurate (beyond a logit gap
    of 60 it takes its direct-evaluation path; scale 400 forces that), and the backward must stay finite and as accurate
    as the fp32 CPU evaluation (a sharper softmax amplifies the cancellation in (t_ij - c_i) of its backward)."""
    from oracle import model_oracle as mo
    model, oracle = make_models(4321, 77)
    sd = dict(oracle.state_dict())
    for layer in (0, 5):
        key = f"message_passing_layers.{layer}.message_passing.module.attn_l"
        sd[key] = sd[key] * scale
    oracle.load_state_dict(sd)
    model.load_state_dict(sd)
    oracle64 = copy.deepcopy(oracle).double()
    n, B = 19, 2
    N = n * (n - 1) // 2
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    t = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    G = mo.batch_line_graphs(n, B)
    y32, _, g32, _ = mo.train_step_reference(oracle, G, x, t)
    y64, _, g64, _ = mo.train_step_reference(oracle64, G, x.double(), t.double())
    y, loss, grads, _ = hip_step(model, n, B, x, t)
    assert np.isfinite(loss) and all(torch.isfinite(g).all() for g in grads.values())
    own = (y32.double() - y64).abs().max().item()
    assert (y.double() - y64).abs().max().item() <= 3 * own + 1e-5 * y64.abs().max().item()
    m = gradient_error_metrics(grads, g32, g64)
    assert gradient_errors_acceptable(m), m
