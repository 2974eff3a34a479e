"""GPU tests of the batched end-to-end path (gnngls_amd.pipeline) and size-independent properties of the forward
at sizes where the CPU oracle is too slow (TSP100 x many, TSP200)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def fbits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.uint64)


@pytest.fixture(scope="module")
def model():
    from gnngls_amd import pipeline
    return pipeline.synthetic_model(seed=7)


def test_solve_batch_matches_oracle_chain(model):
    """features -> forward -> regret -> nearest_neighbor -> tour_cost -> guided_local_search, K outer iterations:
    the search part is bit-exact against the CPU oracle fed with the GPU's own regret predictions."""
    from gnngls_amd import pipeline
    from gnngls_amd.synthetic import random_instances
    from oracle import gls_oracle as go
    n, B, K = 30, 6, 4
    D_host, _ = random_instances(np.random.default_rng(3), B, n)
    D = torch.from_numpy(D_host).cuda()
    sc = pipeline.Scalers.fit_weights(D)
    r = pipeline.solve_batch(D, model, sc, guides=("regret_pred", "weight"), max_outer_iters=K, perturbation_moves=20,
                             trace_cap=4096, keep_regret=True)
    R = r.regret_pred.cpu().numpy()
    assert (R >= 0).all() and np.array_equal(R, R.transpose(0, 2, 1))
    for b in range(B):
        init = go.nearest_neighbor(R[b])
        cost = go.tour_cost(init, D_host[b])
        assert fbits(r.init_cost[b].item()) == fbits(cost)
        o = go.guided_local_search(D_host[b], np.stack([R[b], D_host[b]]), init, cost, perturbation_moves=20,
                                   max_outer_iters=K)
        L = o["trace_len"]
        assert int(r.moves[b]) == L
        assert np.array_equal(fbits(r.trace_cost[b, :L].cpu().numpy()), fbits(o["trace"]))
        assert r.best_tour[b].cpu().tolist() == o["best_tour"]
        assert fbits(r.best_cost[b].item()) == fbits(o["best_cost"])


def test_solve_batch_start_tour_rule_with_weight_first(model):
    """guides = ['weight', 'regret_pred'] (the alternating-guide case algorithms.py:147 supports): the reference starts
    from nearest_neighbor on 'regret_pred' whenever that guide is used at all (test.py:70-88), not on the first guide.
    The oracle is driven exactly as test.py drives the reference."""
    from gnngls_amd import pipeline
    from gnngls_amd.synthetic import random_instances
    from oracle import gls_oracle as go
    n, B, K = 24, 5, 6
    D_host, _ = random_instances(np.random.default_rng(17), B, n)
    D = torch.from_numpy(D_host).cuda()
    sc = pipeline.Scalers.fit_weights(D)
    r = pipeline.solve_batch(D, model, sc, guides=("weight", "regret_pred"), max_outer_iters=K, perturbation_moves=20,
                             trace_cap=4096, keep_regret=True)
    R = r.regret_pred.cpu().numpy()
    differs = 0
    for b in range(B):
        init = go.nearest_neighbor(R[b])                                     # test.py:85
        differs += init != go.nearest_neighbor(D_host[b])
        cost = go.tour_cost(init, D_host[b])                                 # test.py:90
        assert fbits(r.init_cost[b].item()) == fbits(cost)
        o = go.guided_local_search(D_host[b], np.stack([D_host[b], R[b]]), init, cost, perturbation_moves=20,
                                   max_outer_iters=K)
        L = o["trace_len"]
        assert int(r.moves[b]) == L
        assert np.array_equal(fbits(r.trace_cost[b, :L].cpu().numpy()), fbits(o["trace"]))
        assert r.best_tour[b].cpu().tolist() == o["best_tour"]
    assert differs > 0                                                       # the rule is observable on this batch


@pytest.mark.parametrize("n,B,limit", [(50, 128, 1.5), (100, 1024, 2.0), (200, 256, 4.0)])
def test_solve_batch_config_sizes_end_to_end(model, n, B, limit):
    """BASELINE configs[1] (TSP50 x 128: GNN forward + GLS), configs[2] (TSP100 x 1024, the headline shape: forward over
    5.07 million rows, then all 1024 searches resident) and configs[4] (TSP200 x 256 per GPU) through the whole
    pipeline in ONE round: forward, regret guide, start tours, search within the remaining budget; results are valid
    tours with consistent costs, never worse than the start, no aborts, the budget is respected."""
    import time
    from gnngls_amd import ops, pipeline
    from gnngls_amd.synthetic import random_instances
    D = torch.from_numpy(random_instances(np.random.default_rng(n), B, n)[0]).cuda()
    sc = pipeline.Scalers.fit_weights(D)
    pipeline.solve_batch(D[:2].contiguous(), model, sc, time_limit=0.05)     # warm-up (module load, workspace)
    t0 = time.time()
    r = pipeline.solve_batch(D, model, sc, guides=("regret_pred",), time_limit=limit, imp_cap=64)
    wall = time.time() - t0
    assert r.timing["chunks"] == 1 and wall < limit + 1.5
    assert r.timing["forward_s"] < limit / 2
    assert (r.status == 0).all() and (r.outer_iters > 5).all()
    assert (r.best_cost <= r.init_cost).all()
    bt = r.best_tour.cpu().numpy()
    assert (bt[:, 0] == 0).all() and (bt[:, -1] == 0).all()
    assert (np.sort(bt[:, :-1], axis=1) == np.arange(n)[None]).all()
    assert torch.allclose(ops.tour_cost(r.best_tour, D), r.best_cost, rtol=1e-12, atol=0)
    L = r.imp_len.cpu().numpy()
    last = r.imp_cost.cpu().numpy()[np.arange(B), np.minimum(L, 64) - 1]
    assert np.array_equal(fbits(last), fbits(r.best_cost.cpu().numpy()))


def test_solve_batch_chunking_and_weight_guide(model):
    """A batch larger than the chunk size is processed in chunks with identical results; guides=['weight'] needs no model."""
    from gnngls_amd import pipeline
    from gnngls_amd.synthetic import random_instances
    n, B = 20, 10
    D = torch.from_numpy(random_instances(np.random.default_rng(5), B, n)[0]).cuda()
    a = pipeline.solve_batch(D, guides=("weight",), max_outer_iters=5, chunk=4)
    b = pipeline.solve_batch(D, guides=("weight",), max_outer_iters=5, chunk=64)
    assert a.timing["chunks"] == 3 and b.timing["chunks"] == 1
    assert torch.equal(a.best_tour, b.best_tour) and torch.equal(a.best_cost, b.best_cost)
    with pytest.raises(ValueError):
        pipeline.solve_batch(D, guides=("regret_pred",))
    with pytest.raises(ValueError):
        pipeline.solve_batch(D, guides=("width",))
    with pytest.raises(ValueError):
        pipeline.solve_batch(D, guides=("weight",), budget="per_gpu")


def test_budget_policies():
    """per_instance: every round gets the full time limit (test.py:64,92); per_batch: the rounds share it."""
    import time
    from gnngls_amd import pipeline
    from gnngls_amd.synthetic import random_instances
    n, B = 20, 12
    D = torch.from_numpy(random_instances(np.random.default_rng(6), B, n)[0]).cuda()
    pipeline.solve_batch(D[:4], guides=("weight",), time_limit=0.05, chunk=4)          # warm-up (module load)
    t0 = time.time()
    a = pipeline.solve_batch(D, guides=("weight",), time_limit=0.6, chunk=4, budget="per_instance")
    t1 = time.time()
    b = pipeline.solve_batch(D, guides=("weight",), time_limit=0.6, chunk=4, budget="per_batch")
    t2 = time.time()
    assert a.timing["chunks"] == b.timing["chunks"] == 3
    assert 1.8 <= t1 - t0 < 2.6 and 0.55 <= t2 - t1 < 1.2
    assert b.outer_iters.float().mean() < 0.6 * a.outer_iters.float().mean()
    assert (b.best_cost <= b.init_cost).all() and (a.best_cost <= b.best_cost + 1e-9).float().mean() > 0.5


@pytest.mark.parametrize("n,B", [(100, 8), (200, 2)])
def test_forward_relabelling_equivariance(model, n, B):
    """Size-independent property of the GNN on the line graph of K_n: relabelling the TSP nodes with a permutation
    permutes the predicted regret matrix the same way (R'[p[i], p[j]] = R[i, j]); the instances of a batch do not
    interact (evaluating an instance alone gives the same bits)."""
    from gnngls_amd import pipeline
    from gnngls_amd.synthetic import random_instances
    rng = np.random.default_rng(n)
    D_host, _ = random_instances(rng, B, n)
    D = torch.from_numpy(D_host).cuda()
    sc = pipeline.Scalers.fit_weights(D)
    R = pipeline.predict_regret(model, D, sc)
    assert torch.isfinite(R).all() and (R >= 0).all() and torch.equal(R, R.transpose(1, 2))
    perm = torch.from_numpy(rng.permutation(n)).cuda()
    Dp = torch.empty_like(D)
    Dp[:, perm[:, None], perm[None, :]] = D
    Rp = pipeline.predict_regret(model, Dp.contiguous(), sc)
    back = Rp[:, perm[:, None], perm[None, :]]
    scale = R.abs().max().item()
    assert (back - R).abs().max().item() <= 2e-5 * scale + 1e-12        # different summation orders, fp32
    R0 = pipeline.predict_regret(model, D[:1].contiguous(), sc)
    assert torch.equal(R0[0], R[0])
