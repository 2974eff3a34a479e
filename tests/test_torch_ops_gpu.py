"""`torch.ops.gnngls.*` (torch.library registration over the C ABI) against the ctypes path: bitwise equal."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_custom_ops_equal_the_ctypes_path():
    from gnngls_amd import models, ops, pipeline, torch_ops  # noqa: F401
    from gnngls_amd.synthetic import random_instances
    n, B, K = 40, 6, 5
    D = torch.from_numpy(random_instances(np.random.default_rng(8), B, n)[0]).cuda()
    tour = ops.nearest_neighbor(D)
    cost = ops.tour_cost(tour, D)
    assert torch.equal(torch.ops.gnngls.two_opt_delta_all(tour, D).view(torch.int64), ops.two_opt_delta_all(tour, D).view(torch.int64))
    assert torch.equal(torch.ops.gnngls.relocate_delta_all(tour, D).view(torch.int64), ops.relocate_delta_all(tour, D).view(torch.int64))
    ls = ops.gls_run(D, None, tour, cost, max_outer_iters=0)
    t2, c2, m2 = torch.ops.gnngls.local_search(tour, cost, D, False)
    assert torch.equal(t2, ls.best_tour) and torch.equal(c2, ls.best_cost) and torch.equal(m2, ls.trace_len)
    model = pipeline.synthetic_model(seed=3)
    sc = pipeline.Scalers.fit_weights(D)
    feat = models.pack_features(D, sc.feat_scale, sc.feat_min)
    y = torch.ops.gnngls.regret_forward(feat, model.pack_weights("cuda"), n, 8, 16, 512, len(model.message_passing_layers))
    assert torch.equal(y, models.regret_forward(model, feat, B, n))
    R = models.unpack_regret(y, n, sc.regret_scale, sc.regret_min)
    guides = torch.stack([R, D]).contiguous()
    ref = ops.gls_run(D, guides, tour, cost, perturbation_moves=20, max_outer_iters=K, trace_cap=2048)
    bt, bc, it, tr, tl = torch.ops.gnngls.gls_run(D, guides, tour, cost, 20, K, 0.0, False, 2048)
    assert torch.equal(bt, ref.best_tour) and torch.equal(bc.view(torch.int64), ref.best_cost.view(torch.int64))
    assert torch.equal(it, ref.outer_iters) and torch.equal(tl, ref.trace_len)
    assert torch.equal(tr.view(torch.int64), ref.trace_cost.view(torch.int64))
    with pytest.raises(NotImplementedError):                       # no CPU kernel behind the operator
        torch.ops.gnngls.two_opt_delta_all(tour.cpu(), D.cpu())
