"""GPU parity tests of the edge-regret GNN forward (HIP, through the C ABI) against golden outputs
captured from the reference's models.py and against the CPU oracle (oracle/model_oracle.py).
Tolerance: |y - ref| <= 1e-5 * max(|ref|, output scale) with output scale = max|ref| of the instance -- 1e-5
relative on the regret predictions (BASELINE.json north_star) with the absolute floor outputs near zero need.  (Rounds 1-5
asserted the looser sum 1e-5 |ref| + 1e-5 max|ref|, up to twice this bound at the top of the range.)"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-5
ESCAPES = []      # (n, instance, max err, fp32 reference's own err) of cases that passed on the small-graph clause only


def regret_bound(ref):
    """1e-5 * max(|ref|, output scale), output scale = max|ref|"""
    ref = np.abs(np.asarray(ref, dtype=np.float64))
    return RTOL * np.maximum(ref, ref.max())


def assert_regret_close(y, ref):
    y, ref = np.asarray(y, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
    err = np.abs(y - ref)
    bound = regret_bound(ref)
    assert (err <= bound).all(), f"max err {err.max():.3e}, worst ratio {(err / bound).max():.2f}"


def make_models(seed=1234, sd_seed=99):
    from gnngls_amd.models import EdgePropertyPredictionModel
    from oracle import model_oracle as mo
    torch.manual_seed(seed)
    oracle = mo.EdgeRegretModelOracle(1, 128, 1, 3, n_heads=8)
    sd = mo.synthetic_state_dict(oracle, seed=sd_seed)
    oracle.load_state_dict(sd)
    oracle.eval()
    model = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
    missing = model.load_state_dict(sd)          # identical key layout (models.py / SURVEY 8a-a1)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.eval().to("cuda")
    return model, oracle, sd


@pytest.mark.parametrize("n", [5, 10, 20])
def test_forward_golden(n):
    from gnngls_amd.models import LineGraph
    g = np.load(os.path.join(GOLD, f"model_n{n}.npz"))
    model, _, sd = make_models(int(g["model_seed"]), int(g["sd_seed"]))
    checksum = float(sum(v.double().abs().sum() for v in sd.values() if v.dtype.is_floating_point))
    assert checksum == float(g["sd_checksum"])
    G = LineGraph(n).to("cuda")
    with torch.no_grad():
        y = model(G, torch.from_numpy(g["x"]).cuda())
    assert y.shape == g["y"].shape
    assert_regret_close(y.cpu().numpy(), g["y"])


@pytest.mark.parametrize("n,B", [(3, 4), (4, 3), (33, 3), (50, 2), (65, 2), (100, 2), (150, 1)])   # 150: head-split gat_rows
def test_forward_batch_vs_oracle(n, B):
    """HIP fp32 forward vs the CPU oracle on seeded random batches.

    The bar is 1e-5 * max(|ref|, max|ref|) against the oracle evaluated in fp64 --
    the exact value both fp32 implementations approximate.  A plain fp32 evaluation of the
    reference's own graph (the fp32 oracle, different summation order) is itself only within
    3e-6 .. 1.3e-5 of that value (worst for tiny ill-conditioned graphs such as n=4 where the
    output is a small difference of large activations), so for graphs smaller than every BASELINE
    config (n < 20) the HIP result may instead be no further from the exact value than 3x the fp32
    reference's own rounding error.  From n = 20 up the 1e-5 bound is asserted with no alternative.
    ESCAPES collects the cases that needed the alternative (read by scripts/forward_parity_campaign.py)."""
    import copy
    from gnngls_amd.models import LineGraph
    from oracle import model_oracle as mo
    model, oracle, _ = make_models()
    oracle64 = copy.deepcopy(oracle).double()
    N = n * (n - 1) // 2
    rng = np.random.default_rng(n)
    x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    with torch.no_grad():
        y = model(LineGraph(n, batch=B).to("cuda"), x.cuda()).cpu().numpy().reshape(B, N).astype(np.float64)
        G1 = mo.line_graph_networkx(n)
        for b in range(B):
            xb = x[b * N:(b + 1) * N]
            ref64 = oracle64(G1, xb.double()).numpy().reshape(-1)
            ref32 = oracle(G1, xb).numpy().reshape(-1).astype(np.float64)
            err = np.abs(y[b] - ref64)
            bound = regret_bound(ref64)
            ref_err = np.abs(ref32 - ref64).max()
            strict_ok = bool((err <= bound).all())
            if not strict_ok and n < 20:
                ESCAPES.append((n, b, float(err.max()), float(ref_err)))
            assert strict_ok or (n < 20 and err.max() <= 3.0 * ref_err), \
                f"n={n} b={b}: max err {err.max():.3e} (bound {bound.min():.3e}); fp32 reference's own error {ref_err:.3e}"
            # and directly against the fp32 reference path, the form the north star states -- wherever that path is itself
            # a usable yardstick (its own rounding error well inside the bar; not so for some tiny ill-conditioned graphs)
            if n >= 5 and ref_err <= 0.5 * RTOL * np.abs(ref64).max():
                err32 = np.abs(y[b] - ref32)
                assert (err32 <= regret_bound(ref32)).all(), \
                    f"n={n} b={b}: max |hip - fp32 reference| {err32.max():.3e}"


def test_forward_n200_vs_oracle():
    """BASELINE configs[4] size: ONE TSP200 instance (19,900 line-graph nodes, 7.9 million arcs) through the HIP forward
    -- the head-split gat_rows_kernel<4> (two workgroups per TSP row, 115 KB tiles) -- against the oracle evaluated in
    fp64 (arcs from the closed-form rule, aggregation per destination range: tests/test_model_oracle.py pins both to the
    networkx line graph), 1e-5 * max(|ref|, max|ref|), no alternative clause."""
    import copy
    from gnngls_amd.models import LineGraph
    from oracle import model_oracle as mo
    model, oracle, _ = make_models()
    oracle64 = copy.deepcopy(oracle).double()
    n = 200
    N = n * (n - 1) // 2
    x = torch.from_numpy(np.random.default_rng(n).random((N, 1)).astype(np.float32))
    with torch.no_grad():
        y = model(LineGraph(n).to("cuda"), x.cuda()).cpu().numpy().reshape(-1)
        ref = oracle64(mo.line_graph_arcs_closed_form(n), x.double()).numpy().reshape(-1)
    assert np.isfinite(y).all() and y.shape == ref.shape
    assert_regret_close(y, ref)


def test_forward_headline_batch_instances_alone_and_vs_oracle():
    """BASELINE configs[2] shape: 1024 x TSP100 in one call (5.07 million rows, 13.6 GB of workspace).  Instances 0, 511
    and 1023 of the batch are bit for bit what the same instance gives alone (the instances of a batch do not interact,
    wherever they sit in the launch grid), and instance 511 matches the fp64 oracle at 1e-5."""
    import copy
    from gnngls_amd import models as M
    from oracle import model_oracle as mo
    model, oracle, _ = make_models()
    n, B = 100, 1024
    N = n * (n - 1) // 2
    x = torch.from_numpy(np.random.default_rng(1024).random((B * N, 1)).astype(np.float32)).cuda()
    with torch.no_grad():
        y = M.regret_forward(model, x, B, n).reshape(B, N)
        assert torch.isfinite(y).all()
        for b in (0, 511, 1023):
            alone = M.regret_forward(model, x[b * N:(b + 1) * N].contiguous(), 1, n).reshape(N)
            assert torch.equal(alone, y[b]), b
        ref = copy.deepcopy(oracle).double()(mo.line_graph_arcs_closed_form(n), x[511 * N:512 * N].cpu().double())
    assert_regret_close(y[511].cpu().numpy(), ref.numpy().reshape(-1))


def test_forward_headline_batch_eight_instances_vs_oracle_fixtures():
    """The 1,024-instance TSP100 call again, with the rows of the 16 fixture instances (tests/golden/forward_error_n100.npz: fp64
    oracle outputs made on the build container's CPUs) spread over the batch -- instances 0, 64, 129, ... -- for the two checkpoints
    at initialisation scale: every one of their 8 instances within the bar wherever it sits in the launch grid."""
    import sys
    sys.path.insert(0, GOLD)
    import make_forward_error_fixtures as F
    from gnngls_amd import models as M
    fx = np.load(os.path.join(GOLD, "forward_error_n100.npz"))
    n, B = 100, 1024
    N = n * (n - 1) // 2
    per = int(fx["per_checkpoint"])
    for c, (ms, ss, kind) in enumerate(fx["checkpoints"].tolist()):
        if kind != 0:
            continue
        _, sd, _ = F.checkpoint(ms, ss, kind)
        model = M.EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
        model.load_state_dict(sd)
        model.eval().to("cuda")
        x = torch.from_numpy(np.random.default_rng(77 + c).random((B * N, 1)).astype(np.float32))
        slots = [(129 * k + 64 * c) % B for k in range(per)]
        for k, b in enumerate(slots):
            x[b * N:(b + 1) * N] = torch.from_numpy(F.features(n, c, k))
        with torch.no_grad():
            y = M.regret_forward(model, x.cuda(), B, n).reshape(B, N).cpu().numpy()
        for k, b in enumerate(slots):
            assert_regret_close(y[b], fx["ref64"][c, k])


@pytest.mark.parametrize("n", [100, 200])
def test_forward_error_fixtures(n):
    """The sizes the bench runs, 16 instances each over four seeded checkpoints (tests/golden/make_forward_error_fixtures.py): two
    at initialisation scale -- the bar is asserted as it stands -- and two with trained-like weight scales (BatchNorm gamma up to 4,
    calibrated running statistics, GATConv fc at 3 x / 1 x its initial gain).  The latter are ill-conditioned in fp32: the fixtures
    record that a plain fp32 evaluation of the reference's own graph is 1-35 x the bar away from the fp64 value there, so what is
    asserted for them is that the HIP forward is no further from the exact value than 3 x that evaluation (the clause the tiny
    graphs of test_forward_batch_vs_oracle have); scripts/forward_error_fixtures.py prints the table
    (profiles/r06_forward_error_fixtures.txt)."""
    import sys
    sys.path.insert(0, GOLD)
    import make_forward_error_fixtures as F
    from gnngls_amd import models as M
    fx = np.load(os.path.join(GOLD, f"forward_error_n{n}.npz"))
    per = int(fx["per_checkpoint"])
    for c, (ms, ss, kind) in enumerate(fx["checkpoints"].tolist()):
        stats = {k.split(":", 1)[1]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(f"stats{c}:")} or None
        _, sd, _ = F.checkpoint(ms, ss, kind, stats)
        model = M.EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
        model.load_state_dict(sd)
        model.eval().to("cuda")
        x = torch.from_numpy(np.concatenate([F.features(n, c, k) for k in range(per)])).cuda()
        with torch.no_grad():
            y = M.regret_forward(model, x, per, n).cpu().numpy().astype(np.float64)
        for k in range(per):
            ref, own = fx["ref64"][c, k], float(fx["fp32_eval_max_err"][c, k])
            err = np.abs(y[k] - ref)
            if kind == 0:
                assert (err <= regret_bound(ref)).all(), (n, c, k, err.max() / (RTOL * np.abs(ref).max()))
            else:
                assert (err <= regret_bound(ref)).all() or err.max() <= 3.0 * own, (n, c, k, err.max(), own)


@pytest.mark.parametrize("in_dim,n,B", [(3, 12, 2), (2, 33, 1), (40, 8, 1)])   # (40 > 32: embed and the first fc as two launches)
def test_forward_multi_feature_input(in_dim, n, B):
    """Input width other than 1 (test.py:41 takes it from the dataset; the reference's feature sets are built by
    datasets.py:14-34): embed layer [128, in_dim], same 1e-5 bar against the fp64 oracle."""
    import copy
    from gnngls_amd.models import EdgePropertyPredictionModel, LineGraph
    from oracle import model_oracle as mo
    torch.manual_seed(20 + in_dim)
    oracle = mo.EdgeRegretModelOracle(in_dim, 128, 1, 3, n_heads=8)
    sd = mo.synthetic_state_dict(oracle, seed=7)
    oracle.load_state_dict(sd)
    model = EdgePropertyPredictionModel(in_dim, 128, 1, 3, n_heads=8)
    model.load_state_dict(sd)
    model.eval().to("cuda")
    N = n * (n - 1) // 2
    x = torch.from_numpy(np.random.default_rng(n).random((B * N, in_dim)).astype(np.float32))
    o64 = copy.deepcopy(oracle).double().eval()
    with torch.no_grad():
        y = model(LineGraph(n, batch=B).to("cuda"), x.cuda()).cpu().numpy().reshape(B, N)
        for b in range(B):
            ref = o64(mo.line_graph_networkx(n), x[b * N:(b + 1) * N].double()).numpy().reshape(-1)
            assert_regret_close(y[b], ref)


def test_forward_small_workspace_chunks():
    """A workspace that holds one instance at a time gives the same result as the full batch."""
    from gnngls_amd import models as M
    model, _, _ = make_models()
    n, B = 20, 5
    N = n * (n - 1) // 2
    x = torch.rand(B * N, 1, device="cuda")
    y_full = M.regret_forward(model, x, B, n)
    model._workspace = None
    y_chunk = M.regret_forward(model, x, B, n, max_workspace_bytes=1)
    assert torch.equal(y_full, y_chunk)


def test_gatconv_bias_key_is_accepted():
    """DGL >= 0.7 checkpoints carry message_passing.module.bias; it is folded into BN1's shift."""
    from gnngls_amd.models import EdgePropertyPredictionModel, LineGraph
    model, oracle, sd = make_models()
    sd = dict(sd)
    bias = 0.05 * torch.randn(128)
    sd["message_passing_layers.0.message_passing.module.bias"] = bias
    m2 = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
    m2.load_state_dict(sd)
    m2.eval().to("cuda")
    n = 8
    G = LineGraph(n).to("cuda")
    x = torch.rand(G.number_of_nodes(), 1)
    with torch.no_grad():
        y0 = model(G, x.cuda())
        y1 = m2(G, x.cuda())
    assert not torch.allclose(y0, y1)


def test_pack_unpack_scalers():
    """gnngls_pack_features / gnngls_unpack_regret reproduce sklearn's fp32 MinMaxScaler arithmetic."""
    from gnngls_amd import models as M
    g = np.load(os.path.join(GOLD, "misc.npz"))
    s, m = float(g["scaler_scale"][0]), float(g["scaler_min"][0])
    n = 30
    W = g["nn_W_weight"]
    feat = M.pack_features(torch.from_numpy(W[None]).cuda(), s, m).cpu().numpy()[0]
    assert np.array_equal(feat, g["scaler_fwd"].reshape(-1))       # golden rows are in G.edges (combinations) order
    y = torch.from_numpy(g["scaler_inv_in"].reshape(1, -1)).cuda()
    R = M.unpack_regret(y, n, s, m).cpu().numpy()[0]
    iu = np.triu_indices(n, 1)
    expect = np.maximum(g["scaler_inv"].reshape(-1).astype(np.float64), 0)
    assert np.array_equal(R[iu], expect) and np.array_equal(R, R.T) and (np.diag(R) == 0).all()


@pytest.mark.parametrize("scale", [8.0, 300.0])
def test_forward_with_saturated_attention(scale):
    """Large attention logits: the factorised softmax weights of gat_rows_kernel (exp(el - M) * exp(er + M - max)) must
    neither overflow nor lose the result, and beyond a logit gap of 60 the kernel takes its direct-evaluation path."""
    import copy
    from gnngls_amd.models import EdgePropertyPredictionModel, LineGraph
    from oracle import model_oracle as mo
    _, oracle, sd = make_models()
    sd = dict(sd)
    for layer in (0, 3):
        sd[f"message_passing_layers.{layer}.message_passing.module.attn_l"] = \
            sd[f"message_passing_layers.{layer}.message_passing.module.attn_l"] * scale
    oracle.load_state_dict(sd)
    oracle64 = copy.deepcopy(oracle).double().eval()
    model = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
    model.load_state_dict(sd)
    model.eval().to("cuda")
    n = 23
    G = mo.line_graph_networkx(n)
    x = torch.rand(G.number_of_nodes(), 1)
    with torch.no_grad():
        y = model(LineGraph(n).to("cuda"), x.cuda()).cpu().double().numpy().reshape(-1)
        ref64 = oracle64(G, x.double()).numpy().reshape(-1)
        ref32 = oracle.eval()(G, x).double().numpy().reshape(-1)
    assert np.isfinite(y).all()
    err, own = np.abs(y - ref64), np.abs(ref32 - ref64).max()
    assert (err <= regret_bound(ref64)).all() or err.max() <= 3.0 * own, (err.max(), own)


def test_feed_forward_block_paths_agree_and_fc_rides_in_the_block():
    """The inference forward runs the feed-forward block on the bf16 matrix pipe (three bf16 pieces per fp32 operand, six products)
    with the next layer's fc (models.py:23) chained into the same launch, the FIRST layer's fc folded into the embedding pass (both are
    linear in the input features) and the decision layer (models.py:69) into the last block: no gemm_fc and no decision launch.  The
    fp32 kernel (GNNGLS_FFN_FP32=1: read once per process, hence a child process) must give the same regret predictions to the
    parity bar -- both approximate the same fp64 value (models.py:26-36,40)."""
    import subprocess
    import sys
    import tempfile
    from gnngls_amd import _lib, pipeline
    from gnngls_amd.synthetic import random_instances
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n, B = 40, 6
    D = torch.from_numpy(random_instances(np.random.default_rng(3), B, n)[0]).cuda()
    model = pipeline.synthetic_model(seed=1234)
    sc = pipeline.Scalers.fit_weights(D)
    _lib.profile_enable(True)
    R = pipeline.predict_regret(model, D, sc)
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    layers = prof["ffn_fused"][1]
    bf16 = os.environ.get("GNNGLS_FFN_FP32", "0") in ("", "0")
    assert layers >= 2 and prof["gemm_fc"][1] == (0 if bf16 else layers) and prof["decision"][1] == (0 if bf16 else 1), prof
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "r.npy")
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r)\n"
                "from gnngls_amd import pipeline\n"
                "from gnngls_amd.synthetic import random_instances\n"
                "D = torch.from_numpy(random_instances(np.random.default_rng(3), %d, %d)[0]).cuda()\n"
                "R = pipeline.predict_regret(pipeline.synthetic_model(seed=1234), D, pipeline.Scalers.fit_weights(D))\n"
                "np.save(%r, R.cpu().numpy())\n" % (root, B, n, out))
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, GNNGLS_FFN_FP32="1"), timeout=600)
        ref = np.load(out)
    got = R.cpu().numpy()
    off = ~np.eye(n, dtype=bool)
    assert_regret_close(got[:, off], ref[:, off])
