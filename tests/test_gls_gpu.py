"""GPU parity tests of the guided-local-search HIP path (through the C ABI) against
(a) golden vectors captured from the reference and (b) the CPU oracle on seeded random inputs.
Bit-exact: fp64 costs/deltas compared by bit pattern, tours and moves by value."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64)).view(np.uint64)


def assert_bits(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    both_nan = np.isnan(a) & np.isnan(b)
    assert np.array_equal(bits(a)[~both_nan], bits(b)[~both_nan])


@pytest.fixture(scope="module")
def ops():
    from gnngls_amd import ops as o
    return o


def dev(x, dtype):
    return torch.as_tensor(np.ascontiguousarray(x)).to(dtype).cuda().contiguous()


def random_instances(rng, B, n):
    pos = rng.random((B, n, 2))
    d = pos[:, :, None, :] - pos[:, None, :, :]
    D = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1])
    D = np.triu(D, 1)
    D = D + D.transpose(0, 2, 1)
    tours = np.zeros((B, n + 1), dtype=np.int32)
    for b in range(B):
        tours[b, 1:n] = 1 + rng.permutation(n - 1)
    return D, tours


def check_ops_case(ops, g, prefix=""):
    tour, D = g[prefix + "tour"], g[prefix + "D"]
    n = len(tour) - 1
    t, d = dev(tour[None], torch.int32), dev(D[None], torch.float64)
    assert_bits(ops.two_opt_delta_all(t, d)[0].cpu().numpy(), g[prefix + "two_opt_table"])
    assert_bits(ops.relocate_delta_all(t, d)[0].cpu().numpy(), g[prefix + "relocate_table"])
    for fi in (0, 1):
        for op, name in ((0, "two_opt"), (1, "relocate")):
            delta, move, nt = ops.best_move(t, d, op, None, bool(fi))
            assert_bits(delta.cpu().numpy()[0], g[f"{prefix}{name}_a2a_fi{fi}_delta"])
            assert nt[0].cpu().tolist() == g[f"{prefix}{name}_a2a_fi{fi}_tour"].tolist()
            # all o2a positions as one batch
            B = n - 1
            tb = t.expand(B, -1).contiguous()
            db = d.expand(B, -1, -1).contiguous()
            pos = torch.arange(1, n, dtype=torch.int32, device="cuda")
            delta, move, nt = ops.best_move(tb, db, op, pos, bool(fi))
            assert_bits(delta.cpu().numpy(), g[f"{prefix}{name}_o2a_fi{fi}_delta"])
            assert np.array_equal(nt.cpu().numpy(), g[f"{prefix}{name}_o2a_fi{fi}_tour"])


@pytest.mark.parametrize("n", [5, 8, 20, 50, 100])
def test_operators_golden(ops, n):
    check_ops_case(ops, np.load(os.path.join(GOLD, f"ops_n{n}.npz")))


def test_operators_ties_and_isclose_golden(ops):
    g = np.load(os.path.join(GOLD, "ops_ties.npz"))
    for c in range(int(g["n_cases"])):
        check_ops_case(ops, g, prefix=f"c{c}_")


@pytest.mark.parametrize("n", [8, 20, 50, 100])
def test_local_search_golden(ops, n):
    g = np.load(os.path.join(GOLD, f"ls_n{n}.npz"))
    d = dev(g["D"][None], torch.float64)
    for fi in (0, 1):
        r = ops.gls_run(d, None, dev(g[f"fi{fi}_init_tour"][None], torch.int32),
                        dev(np.array([g[f"fi{fi}_init_cost"]]), torch.float64),
                        first_improvement=bool(fi), max_outer_iters=0, trace_cap=4096)
        L = int(r.trace_len[0])
        assert r.best_tour[0].cpu().tolist() == g[f"fi{fi}_tour"].tolist()
        assert_bits(r.best_cost.cpu().numpy()[0], g[f"fi{fi}_cost"])
        assert_bits(r.trace_cost[0, :L].cpu().numpy(), g[f"fi{fi}_trace"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "gls_c*.npz"))), ids=os.path.basename)
@pytest.mark.parametrize("trace", [True, False])
def test_guided_local_search_golden(ops, path, trace):
    g = np.load(path)
    d = dev(g["D"][None], torch.float64)
    guides = dev(g["guides"][:, None], torch.float64)
    r = ops.gls_run(d, guides, dev(g["init_tour"][None], torch.int32), dev(np.array([g["init_cost"]]), torch.float64),
                    perturbation_moves=int(g["perturbation_moves"]), first_improvement=bool(g["first_improvement"]),
                    max_outer_iters=int(g["K"]), trace_cap=8192 if trace else 0, want_penalty=True)
    assert int(r.status[0]) == 0
    assert int(r.outer_iters[0]) == int(g["K"])
    assert int(r.trace_len[0]) == len(g["trace"])
    if trace:
        assert_bits(r.trace_cost[0, :len(g["trace"])].cpu().numpy(), g["trace"])
    assert r.best_tour[0].cpu().tolist() == g["best_tour"].tolist()
    assert_bits(r.best_cost.cpu().numpy()[0], g["best_cost"])
    assert np.array_equal(r.penalty[0].cpu().numpy(), g["penalty"])


def test_misc_golden(ops):
    g = np.load(os.path.join(GOLD, "misc.npz"))
    for key in ("weight", "regret"):
        t = ops.nearest_neighbor(dev(g[f"nn_W_{key}"][None], torch.float64))
        assert t[0].cpu().tolist() == g[f"nn_tour_{key}"].tolist()
    c = ops.tour_cost(dev(g["tc_tour"][None], torch.int32), dev(g["nn_W_weight"][None], torch.float64))
    assert_bits(c.cpu().numpy()[0], g["tc_cost"])


VARIANTS = {"serial": (0, 1), "team": (1, 1), "fullscan": (0, 0)}     # (team form of the perturbation phase, pruned descent scans)


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("n,B,K", [(7, 16, 6), (17, 8, 8), (20, 32, 10), (33, 16, 6), (64, 8, 4), (65, 8, 4), (100, 16, 3), (130, 4, 2)])
def test_gls_batch_vs_oracle(ops, n, B, K, variant):
    """Seeded random batches: HIP path vs the CPU oracle, one instance per workgroup; perturbation phase on wavefront 0
    (serial) and on all wavefronts of the workgroup (team)."""
    team, prune = VARIANTS[variant]
    if not prune and n < 80:
        pytest.skip("the pruned descent scans only exist for n >= 80: same kernel as the serial case")
    from oracle import gls_oracle as go
    rng = np.random.default_rng(1000 + n)
    D, _ = random_instances(rng, B, n)
    guide = np.maximum(rng.normal(0.05, 0.1, size=D.shape).astype(np.float32).astype(np.float64), 0)
    guide = np.triu(guide, 1)
    guide = guide + guide.transpose(0, 2, 1)
    guides = np.stack([guide, D])
    d, gd = dev(D, torch.float64), dev(guides, torch.float64)
    init = ops.nearest_neighbor(gd[0])
    cost = ops.tour_cost(init, d)
    with ops.gls_team_mode(team), ops.gls_prune_mode(prune):
        assert ops.gls_describe_config(n, B)["team"] == bool(team)
        r = ops.gls_run(d, gd, init, cost, perturbation_moves=20, max_outer_iters=K, trace_cap=1 << 14, want_penalty=True)
    init_h, cost_h = init.cpu().numpy(), cost.cpu().numpy()
    for b in range(B):
        assert init_h[b].tolist() == go.nearest_neighbor(guide[b])
        assert_bits(cost_h[b], go.tour_cost(init_h[b], D[b]))
        o = go.guided_local_search(D[b], guides[:, b], init_h[b], cost_h[b], perturbation_moves=20, max_outer_iters=K)
        L = o["trace_len"]
        assert int(r.trace_len[b]) == L
        assert_bits(r.trace_cost[b, :L].cpu().numpy(), o["trace"])
        assert r.best_tour[b].cpu().tolist() == o["best_tour"]
        assert_bits(r.best_cost[b].item(), o["best_cost"])
        assert np.array_equal(r.penalty[b].cpu().numpy(), o["penalty"])
        assert int(r.evals[b]) == o["evals"]


@pytest.mark.parametrize("n,B,K", [(33, 8, 6), (64, 8, 4), (100, 16, 4), (200, 4, 2)])
def test_executed_evaluation_count(ops, n, B, K):
    """Measurement hook gnngls_profile_set_executed_evals (bench.py's roofline; counting instantiations of the compact-store
    kernel): executed = reference-equivalent evaluations where no scan is pruned (n < 80, or the pruned scans switched off);
    with the pruned scans (2-opt from n = 80, relocate from n = 128) strictly fewer -- and the same search result."""
    rng = np.random.default_rng(4200 + n)
    D, _ = random_instances(rng, B, n)
    d = dev(D, torch.float64)
    g = d[None].contiguous()
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    plain = ops.gls_run(d, g, init, cost, perturbation_moves=20, max_outer_iters=K, penalty_bits=-2)
    out = {}
    for prune in (1, 0):
        with ops.gls_prune_mode(prune), ops.executed_evals(B + 3) as x:
            r = ops.gls_run(d, g, init, cost, perturbation_moves=20, max_outer_iters=K, penalty_bits=-2)     # compact store
        torch.cuda.synchronize()
        out[prune] = (r, x.counts.cpu().numpy())
    (r1, x1), (r0, x0) = out[1], out[0]
    ref = r0.evals.cpu().numpy()
    for r in (r0, r1):                                                             # counting changes no result
        assert torch.equal(r.best_tour, plain.best_tour) and torch.equal(r.evals, plain.evals)
        assert_bits(r.best_cost.cpu().numpy(), plain.best_cost.cpu().numpy())
    assert np.array_equal(x0[:B], ref) and (x0[B:] == 0).all()                 # full scans: every move evaluated once
    if n < 80:
        assert np.array_equal(x1[:B], ref)
    else:
        assert (x1[:B] > 0).all() and (x1[:B] < ref).all()
        if n >= 128:                                                              # both descent scans pruned
            assert (x1[:B] < 0.5 * ref).all()
        # a pruning run on a store without the counting instantiations says so instead of guessing
        with ops.executed_evals(B) as x:
            ops.gls_run(d, g, init, cost, perturbation_moves=20, max_outer_iters=1, penalty_bits=32)
        if ops.gls_describe_config(n, B, 32)["store"] == "lds-tri-i32":
            assert (x.counts.cpu().numpy() == -1).all()


@pytest.mark.parametrize("n,B,K,bits,store,per_cu", [
    (131, 3, 2, -2, "compact", 2), (131, 3, 2, 0, "lds-tri-i32", 1), (160, 3, 2, -2, "compact", 1),
    (160, 3, 2, 0, "lds-tri-i32", 1), (200, 3, 2, 0, "compact", 1), (200, 2, 1, -2, "compact", 1)])
@pytest.mark.parametrize("fi", [False, True])
@pytest.mark.parametrize("team,prune", [(0, 1), (-1, 1), (0, 0)], ids=["serial", "policy", "fullscan"])
def test_gls_large_n_lds_stores_vs_oracle(ops, n, B, K, bits, store, per_cu, fi, team, prune):
    """BASELINE configs[4] regime (TSP200; n = 131..200): one or two workgroups per CU (up to 159 KB of LDS for the
    distance triangle), no row-on-the-lane descent (n - 1 > 128), four register passes of cached utilities -- on the
    store gnngls_gls_run picks by itself and on the compact store (the one TSP200 x 256 runs on).  Two guides, both
    improvement modes: every accepted move, the best tour / cost, the final penalties, the evaluation count and the
    improvement record equal the CPU oracle's bit for bit."""
    from oracle import gls_oracle as go
    cfg = ops.gls_describe_config(n, B, penalty_bits=bits, first_improvement=fi)
    assert cfg["store"] == store and cfg["per_cu"] == per_cu and cfg["lds_bytes"] > 64 * 1024, cfg
    # 16-wave workgroup that owns its CU, B <= number of CUs; round 5: first-improvement runs only (best improvement runs the
    # serial phase in its edge form, which is faster there too)
    policy_team = cfg["store"] == "compact" and cfg["threads"] == 1024 and fi
    assert cfg["team"] == policy_team
    c200 = ops.gls_describe_config(200, 256, first_improvement=True)
    assert c200["store"] == "compact" and c200["team"] and c200["threads"] == 1024 and c200["lds_bytes"] <= 160 * 1024
    assert not ops.gls_describe_config(200, 256)["team"] and ops.gls_describe_run(200, 256)["edge_form"]
    rng = np.random.default_rng(4000 + n)
    D, _ = random_instances(rng, B, n)
    guide = np.maximum(rng.normal(0.05, 0.1, size=D.shape).astype(np.float32).astype(np.float64), 0)
    guide = np.triu(guide, 1)
    guide = guide + guide.transpose(0, 2, 1)
    guides = np.stack([guide, D])
    d, gd = dev(D, torch.float64), dev(guides, torch.float64)
    init = ops.nearest_neighbor(gd[0])
    cost = ops.tour_cost(init, d)
    with ops.gls_team_mode(team), ops.gls_prune_mode(prune):
        assert ops.gls_describe_config(n, B, penalty_bits=bits, first_improvement=fi)["team"] == (team != 0 and policy_team)
        r = ops.gls_run(d, gd, init, cost, perturbation_moves=20, first_improvement=fi, max_outer_iters=K,
                        trace_cap=1 << 14, want_penalty=True, imp_cap=32, penalty_bits=bits)
        plain = ops.gls_run(d, gd, init, cost, perturbation_moves=20, first_improvement=fi, max_outer_iters=K,
                            penalty_bits=bits)                                                              # no-trace kernel
    init_h, cost_h = init.cpu().numpy(), cost.cpu().numpy()
    for b in range(B):
        assert init_h[b].tolist() == go.nearest_neighbor(guide[b])
        o = go.guided_local_search(D[b], guides[:, b], init_h[b], cost_h[b], perturbation_moves=20, first_improvement=fi,
                                   max_outer_iters=K)
        L = o["trace_len"]
        assert int(r.status[b]) == 0 and int(r.trace_len[b]) == L and int(plain.trace_len[b]) == L
        assert_bits(r.trace_cost[b, :L].cpu().numpy(), o["trace"])
        assert r.best_tour[b].cpu().tolist() == o["best_tour"] == plain.best_tour[b].cpu().tolist()
        assert_bits(r.best_cost[b].item(), o["best_cost"])
        assert_bits(plain.best_cost[b].item(), o["best_cost"])
        assert np.array_equal(r.penalty[b].cpu().numpy(), o["penalty"])
        assert int(r.evals[b]) == o["evals"] == int(plain.evals[b])
        assert int(r.imp_len[b]) == o["imp_len"]
        assert_bits(r.imp_cost[b, :o["imp_len"]].cpu().numpy(), o["imp_cost"])


def test_gls_halved_workgroups_with_prune_lists_vs_oracle(ops):
    """n = 84 with more instances than 4-wave workgroups keep resident: the launcher halves the workgroup to 128 threads;
    a lane would then need more than the four list registers the pruned scans keep per lane (8 (n - 1) rows x lanes over
    128 threads = 6 passes), so such a launch runs the full scans although the neighbour lists were built.  Sampled
    instances against the oracle, bit for bit."""
    from oracle import gls_oracle as go
    n, B, K = 84, 1100, 2
    cfg = ops.gls_describe_config(n, B)
    assert cfg["store"] == "compact" and cfg["threads"] == 128
    rng = np.random.default_rng(84)
    D, _ = random_instances(rng, B, n)
    d = dev(D, torch.float64)
    gd = d[None].contiguous()
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    r = ops.gls_run(d, gd, init, cost, perturbation_moves=20, max_outer_iters=K, trace_cap=0)
    assert (r.status == 0).all()
    init_h, cost_h = init.cpu().numpy(), cost.cpu().numpy()
    for b in (0, 1, 511, 1024, 1099):
        o = go.guided_local_search(D[b], D[b][None], init_h[b], cost_h[b], perturbation_moves=20, max_outer_iters=K)
        assert r.best_tour[b].cpu().tolist() == o["best_tour"]
        assert_bits(r.best_cost[b].item(), o["best_cost"])
        assert int(r.evals[b]) == o["evals"]


def test_gls_global_store_fallback(ops):
    """n too large for the LDS triangles -> global-memory store path; same results as the oracle."""
    from oracle import gls_oracle as go
    n, B, K = 210, 2, 1
    assert ops.gls_resident_capacity(n) == 0
    rng = np.random.default_rng(7)
    D, tours = random_instances(rng, B, n)
    d = dev(D, torch.float64)
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    r = ops.gls_run(d, d[None].contiguous(), init, cost, perturbation_moves=10, max_outer_iters=K, trace_cap=1 << 14)
    for b in range(B):
        o = go.guided_local_search(D[b], D[b][None], init[b].cpu().numpy(), cost[b].item(), perturbation_moves=10,
                                   max_outer_iters=K)
        assert int(r.trace_len[b]) == o["trace_len"]
        assert_bits(r.trace_cost[b, :o["trace_len"]].cpu().numpy(), o["trace"])
        assert r.best_tour[b].cpu().tolist() == o["best_tour"]


def test_gls_time_mode_and_properties(ops):
    """Wall-clock mode at TSP100: valid tours, cost == tour_cost(best_tour), never worse than the
    local-search-only result, and the deterministic K-iteration trace is a prefix of a longer run."""
    n, B = 100, 64
    rng = np.random.default_rng(5)
    D, _ = random_instances(rng, B, n)
    d = dev(D, torch.float64)
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    g = d[None].contiguous()
    ls = ops.gls_run(d, None, init, cost, max_outer_iters=0)
    r = ops.gls_run(d, g, init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=0.5)
    torch.cuda.synchronize()
    assert (r.status == 0).all()
    assert (r.outer_iters > 0).all()
    assert (r.best_cost <= ls.best_cost).all()
    bt = r.best_tour.cpu().numpy()
    assert (bt[:, 0] == 0).all() and (bt[:, -1] == 0).all()
    assert all(sorted(row[:-1].tolist()) == list(range(n)) for row in bt)
    rc = ops.tour_cost(r.best_tour, d)
    assert torch.allclose(rc, r.best_cost, rtol=1e-12, atol=1e-12)
    a = ops.gls_run(d, g, init, cost, perturbation_moves=20, max_outer_iters=3, trace_cap=4096)
    b2 = ops.gls_run(d, g, init, cost, perturbation_moves=20, max_outer_iters=6, trace_cap=4096)
    for b in range(B):
        La = int(a.trace_len[b])
        assert int(b2.trace_len[b]) >= La
        assert torch.equal(a.trace_cost[b, :La], b2.trace_cost[b, :La])


@pytest.mark.parametrize("bits", [0, 16, 32])
def test_gls_penalty_width_variants(ops, bits):
    """Both LDS penalty widths give the reference's results (golden, TSP100)."""
    g = np.load(os.path.join(GOLD, "gls_c7_n100_K20.npz"))
    r = ops.gls_run(dev(g["D"][None], torch.float64), dev(g["guides"][:, None], torch.float64),
                    dev(g["init_tour"][None], torch.int32), dev(np.array([g["init_cost"]]), torch.float64),
                    perturbation_moves=int(g["perturbation_moves"]), max_outer_iters=int(g["K"]), trace_cap=8192,
                    want_penalty=True, penalty_bits=bits)
    assert int(r.status[0]) == 0
    assert_bits(r.trace_cost[0, :len(g["trace"])].cpu().numpy(), g["trace"])
    assert r.best_tour[0].cpu().tolist() == g["best_tour"].tolist()
    assert np.array_equal(r.penalty[0].cpu().numpy(), g["penalty"])


@pytest.mark.parametrize("bits16", [16])         # uint16 triangle in LDS (the compact store keeps int32 counters in global memory)
def test_gls_penalty16_overflow_is_detected_and_rerun(ops, bits16):
    """A 16-bit penalty counter that would overflow stops the instance with status 2; ops.gls_run reruns
    it with 32-bit counters, so the final result equals the 32-bit run (test hook lowers the limit)."""
    from gnngls_amd import _lib
    g = np.load(os.path.join(GOLD, "gls_c7_n100_K20.npz"))
    assert g["penalty"].max() > 2
    args = (dev(g["D"][None], torch.float64), dev(g["guides"][:, None], torch.float64),
            dev(g["init_tour"][None], torch.int32), dev(np.array([g["init_cost"]]), torch.float64))
    kw = dict(perturbation_moves=int(g["perturbation_moves"]), max_outer_iters=int(g["K"]), trace_cap=8192, want_penalty=True)
    L = _lib.load()
    _lib.check(L.gnngls_debug_set_penalty16_limit(2))
    try:
        raw = ops.gls_run(*args, penalty_bits=bits16, retry_overflow=False, **kw)
        assert int(raw.status[0]) == ops.STATUS_PENALTY_OVERFLOW
        r = ops.gls_run(*args, penalty_bits=bits16, **kw)
    finally:
        _lib.check(L.gnngls_debug_set_penalty16_limit(65535))
    assert int(r.status[0]) == 0
    assert_bits(r.trace_cost[0, :len(g["trace"])].cpu().numpy(), g["trace"])
    assert r.best_tour[0].cpu().tolist() == g["best_tour"].tolist()
    assert np.array_equal(r.penalty[0].cpu().numpy(), g["penalty"])


@pytest.mark.parametrize("n,B,K,fi", [(3, 8, 0, False), (4, 8, 5, False), (5, 8, 5, True), (48, 6, 4, True), (100, 4, 3, True)])
def test_gls_tiny_and_first_improvement_vs_oracle(ops, n, B, K, fi):
    """Smallest legal instances and first_improvement=True.  n=3 has a single tour (no 2-opt move, one relocate
    pair): only the descent is run there -- the perturbation loop of the reference (`while moves <
    perturbation_moves`, algorithms.py:151) can never accept a move on a 3-cycle and does not terminate."""
    from oracle import gls_oracle as go
    rng = np.random.default_rng(77 + n)
    D, _ = random_instances(rng, B, n)
    d = dev(D, torch.float64)
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    r = ops.gls_run(d, d[None].contiguous(), init, cost, perturbation_moves=5, first_improvement=fi, max_outer_iters=K,
                    trace_cap=4096, want_penalty=True)
    for b in range(B):
        o = go.guided_local_search(D[b], D[b][None], init[b].cpu().numpy(), cost[b].item(), perturbation_moves=5,
                                   first_improvement=fi, max_outer_iters=K)
        assert int(r.status[b]) == 0
        assert int(r.trace_len[b]) == o["trace_len"]
        assert_bits(r.trace_cost[b, :o["trace_len"]].cpu().numpy(), o["trace"])
        assert r.best_tour[b].cpu().tolist() == o["best_tour"]
        assert np.array_equal(r.penalty[b].cpu().numpy(), o["penalty"])


def test_gls_long_run_compact_store_vs_oracle(ops):
    """TSP100 on the compact store (penalties in L2): 150 outer iterations, ~4500 accepted moves, penalties in
    the tens -- every accepted move, the best tour and the final penalty matrix equal the CPU oracle's."""
    from oracle import gls_oracle as go
    n, B, K = 100, 2, 150
    assert ops.gls_resident_capacity(n) == 1024
    rng = np.random.default_rng(2024)
    D, _ = random_instances(rng, B, n)
    d = dev(D, torch.float64)
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    r = ops.gls_run(d, d[None].contiguous(), init, cost, perturbation_moves=20, max_outer_iters=K, trace_cap=1 << 15,
                    want_penalty=True)
    for b in range(B):
        o = go.guided_local_search(D[b], D[b][None], init[b].cpu().numpy(), cost[b].item(), perturbation_moves=20,
                                   max_outer_iters=K)
        L = o["trace_len"]
        assert L > 3000 and int(r.trace_len[b]) == L
        assert_bits(r.trace_cost[b, :L].cpu().numpy(), o["trace"])
        assert r.best_tour[b].cpu().tolist() == o["best_tour"]
        assert np.array_equal(r.penalty[b].cpu().numpy(), o["penalty"]) and o["penalty"].max() >= 5


@pytest.mark.parametrize("team", [0, 1], ids=["serial", "team"])
def test_watchdog_stops_an_endless_perturbation_phase(ops, team):
    """`while moves < perturbation_moves` (algorithms.py:151) ignores the clock: with an unreachable move count the phase never
    ends by itself.  The device watchdog is checked inside the phase too (every 64 penalty steps), in both forms: the run
    comes back with status WATCHDOG, valid tours and the cost of the best tour, instead of hanging."""
    import time
    n, B = 150, 3
    D, _ = random_instances(np.random.default_rng(8), B, n)
    d = dev(D, torch.float64)
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    with ops.gls_team_mode(team):
        assert ops.gls_describe_config(n, B, penalty_bits=-2)["team"] == bool(team)
        t0 = time.time()
        r = ops.gls_run(d, d[None].contiguous(), init, cost, perturbation_moves=2 ** 30, max_outer_iters=5, watchdog_s=0.3,
                        penalty_bits=-2)
        torch.cuda.synchronize()
    assert time.time() - t0 < 5.0
    assert (r.status == ops.STATUS_WATCHDOG).all()
    bt = r.best_tour.cpu().numpy()
    assert (np.sort(bt[:, :-1], axis=1) == np.arange(n)[None]).all() and (bt[:, -1] == 0).all()
    assert torch.allclose(ops.tour_cost(r.best_tour, d), r.best_cost, rtol=1e-12, atol=0)


def test_empty_batches_are_noops(ops):
    d = torch.zeros((0, 5, 5), dtype=torch.float64, device="cuda")
    t = torch.zeros((0, 6), dtype=torch.int32, device="cuda")
    assert ops.tour_cost(t, d).shape == (0,)
    assert ops.nearest_neighbor(d).shape == (0, 6)
    assert ops.two_opt_delta_all(t, d).shape == (0, 6, 6)
    r = ops.gls_run(d, None, t, torch.zeros((0,), dtype=torch.float64, device="cuda"), max_outer_iters=0)
    assert r.best_tour.shape == (0, 6)


def test_full_size_batch_properties(ops):
    """BASELINE configs[2] size (TSP100 x 1024, all resident): every result is a valid tour whose recomputed
    cost equals the reported cost, no instance is worse than plain local search, no watchdog aborts."""
    n, B = 100, 1024
    D, _ = random_instances(np.random.default_rng(11), B, n)
    d = dev(D, torch.float64)
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    ls = ops.gls_run(d, None, init, cost, max_outer_iters=0)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    r = ops.gls_run(d, d[None].contiguous(), init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=1.0)
    t1.record(); torch.cuda.synchronize()
    assert t0.elapsed_time(t1) < 1500.0          # one round: all 1024 workgroups were resident together
    assert (r.status == 0).all() and (r.outer_iters > 100).all()
    assert (r.best_cost <= ls.best_cost).all()
    bt = r.best_tour.cpu().numpy()
    assert (bt[:, 0] == 0).all() and (bt[:, -1] == 0).all()
    assert (np.sort(bt[:, :-1], axis=1) == np.arange(n)[None]).all()
    assert torch.allclose(ops.tour_cost(r.best_tour, d), r.best_cost, rtol=1e-12, atol=0)


def test_batch_larger_than_the_device_capacity(ops):
    """gnngls_gls_run takes any B: workgroups beyond the resident capacity (1024 at TSP100) start when a slot frees up and
    get the full time limit from their own start, so 1030 instances take two rounds; every result is valid."""
    n, B, limit = 100, 1030, 0.3
    assert ops.gls_resident_capacity(n) == 1024
    D, _ = random_instances(np.random.default_rng(21), B, n)
    d = dev(D, torch.float64)
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    r = ops.gls_run(d, d[None].contiguous(), init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=limit)
    t1.record(); torch.cuda.synchronize()
    assert 2 * limit * 1000.0 * 0.95 < t0.elapsed_time(t1) < 2 * limit * 1000.0 + 1500.0
    assert (r.status == 0).all() and (r.outer_iters > 50).all() and (r.best_cost < cost).all()
    bt = r.best_tour.cpu().numpy()
    assert (np.sort(bt[:, :-1], axis=1) == np.arange(n)[None]).all() and (bt[:, -1] == 0).all()
    assert torch.allclose(ops.tour_cost(r.best_tour, d), r.best_cost, rtol=1e-12, atol=0)


@pytest.mark.parametrize("n,B,limit,min_iters", [(50, 128, 1.0, 200), (200, 256, 2.0, 20)])
def test_config_size_batch_properties(ops, n, B, limit, min_iters):
    """BASELINE configs[1] (TSP50 x 128) and configs[4] (TSP200 x 256 per GPU) at full size in wall-clock mode: one
    round (everything resident), valid tours whose recomputed cost equals the reported cost, never worse than plain
    local search, no watchdog aborts; sampled instances re-run on the CPU oracle for exactly the iterations they
    completed give the same tour and cost bit for bit."""
    from oracle import gls_oracle as go
    assert ops.gls_resident_capacity(n) >= B
    D, _ = random_instances(np.random.default_rng(50 + n), B, n)
    d = dev(D, torch.float64)
    init = ops.nearest_neighbor(d)
    cost = ops.tour_cost(init, d)
    g = d[None].contiguous()
    ls = ops.gls_run(d, None, init, cost, max_outer_iters=0)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    r = ops.gls_run(d, g, init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=limit)
    t1.record(); torch.cuda.synchronize()
    assert t0.elapsed_time(t1) < limit * 1000.0 + 1500.0
    assert (r.status == 0).all() and (r.outer_iters >= min_iters).all(), r.outer_iters.min()
    assert (r.best_cost <= ls.best_cost).all()
    bt = r.best_tour.cpu().numpy()
    assert (bt[:, 0] == 0).all() and (bt[:, -1] == 0).all()
    assert (np.sort(bt[:, :-1], axis=1) == np.arange(n)[None]).all()
    assert torch.allclose(ops.tour_cost(r.best_tour, d), r.best_cost, rtol=1e-12, atol=0)
    iters, init_h, cost_h = r.outer_iters.cpu().numpy(), init.cpu().numpy(), cost.cpu().numpy()
    order = np.argsort(iters)
    for b in [int(order[0]), int(order[len(order) // 2])]:           # cheapest and median instance (oracle time)
        if float(iters[b]) * n * n > 4e8:                             # keep the oracle re-run to seconds
            continue
        o = go.guided_local_search(D[b], D[b][None], init_h[b], cost_h[b], perturbation_moves=20,
                                   max_outer_iters=int(iters[b]), trace_cap=1, want_penalty=False)
        assert r.best_tour[b].cpu().tolist() == o["best_tour"], (b, int(iters[b]))
        assert_bits(r.best_cost[b].item(), o["best_cost"])


@pytest.mark.parametrize("n,B,limit", [(30, 12, 0.2), (100, 8, 0.3)])
def test_wall_clock_run_equals_the_oracle_at_the_same_iteration_count(n, B, limit):
    """The headline mode (deadline instead of an iteration count, algorithms.py:146): whatever number of outer iterations
    an instance completed before the deadline, its best tour and cost are bit for bit what the CPU oracle returns for
    exactly that many iterations (trace-free throughput path, per-move tour_cost deferred)."""
    from gnngls_amd import ops
    from oracle import gls_oracle as go
    D, _ = random_instances(np.random.default_rng(77 + n), B, n)
    d = dev(D, torch.float64)
    rng = np.random.default_rng(5)
    guide = np.maximum(rng.normal(0.0, 0.1, size=D.shape).astype(np.float32).astype(np.float64), 0)    # zero-heavy, like
    guide = np.triu(guide, 1) + np.transpose(np.triu(guide, 1), (0, 2, 1))                              # clamped regrets
    g = dev(np.stack([guide, D]), torch.float64)
    init = ops.nearest_neighbor(g[0].contiguous())
    cost = ops.tour_cost(init, d)
    r = ops.gls_run(d, g, init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=limit)
    torch.cuda.synchronize()
    iters = r.outer_iters.cpu().numpy()
    assert (r.status == 0).all() and (iters > 10).all()
    init_h, cost_h = init.cpu().numpy(), cost.cpu().numpy()
    for b in range(B):
        o = go.guided_local_search(D[b], np.stack([guide[b], D[b]]), init_h[b], cost_h[b], perturbation_moves=20,
                                   max_outer_iters=int(iters[b]), trace_cap=1, want_penalty=False)
        assert r.best_tour[b].cpu().tolist() == o["best_tour"], (b, int(iters[b]))
        assert_bits(r.best_cost[b].item(), o["best_cost"])


@pytest.mark.parametrize("n,B,K,min_optimal,max_mean_gap", [(14, 32, 200, 0.95, 0.05), (20, 16, 400, 0.75, 0.5)])
def test_search_against_exact_optimum(n, B, K, min_optimal, max_mean_gap):
    """The true optimality gap (test.py:104) on instances small enough for the Held-Karp DP (oracle/held_karp.c): no
    result may beat the optimum, and guided local search closes almost all of them."""
    from gnngls_amd import ops
    from oracle import held_karp
    rng = np.random.default_rng(1000 + n)
    pos = rng.random((B, n, 2))
    D_host = np.linalg.norm(pos[:, :, None] - pos[:, None], axis=3)
    opt = held_karp.optima(D_host, workers=4)
    D = torch.from_numpy(D_host).cuda()
    init = ops.nearest_neighbor(D)
    r = ops.gls_run(D, D[None].contiguous(), init, ops.tour_cost(init, D), perturbation_moves=20, max_outer_iters=K)
    gap = r.best_cost.cpu().numpy() / opt - 1.0
    assert gap.min() >= -1e-12
    assert (gap <= 1e-12).mean() >= min_optimal and gap.mean() * 100 <= max_mean_gap, (gap.mean() * 100, (gap <= 1e-12).mean())


def test_c_abi_from_plain_c(tmp_path):
    """examples/gls_from_c.c: the library driven from C with nothing but the HIP runtime (no Python objects, no torch
    types cross the boundary): compiles with gcc against include/gnngls_hip.h and improves every tour."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "gls_from_c")
    subprocess.check_call(["gcc", "-O2", os.path.join(root, "examples", "gls_from_c.c"), "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-I" + os.path.join(root, "include"), "-L" + os.path.join(root, "gnngls_amd"),
                           "-lgnngls_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
                           "-Wl,-rpath," + os.path.join(root, "gnngls_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.check_output([exe, "40", "32", "50"], text=True)
    assert "0 anomalies" in out and "32 TSP40 instances" in out, out
