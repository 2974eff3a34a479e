"""Randomised parity campaign of the search kernel against the CPU oracle: many sizes, integer-lattice distance
matrices (exact ties, deltas that are exactly 0 or ~1e-8), zero-heavy guides, all storage policies, both
improvement modes.  Every accepted move, the best tour and the final penalties must match bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_case(rng, n, kind):
    if kind == "euclid":
        pos = rng.random((n, 2))
        D = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1))
    elif kind == "lattice":                       # many exact ties and zero deltas
        pos = rng.integers(0, 6, size=(n, 2)).astype(np.float64)
        D = np.abs(pos[:, None] - pos[None]).sum(-1) + 1.0
    elif kind == "nonmetric":                     # round 5: uniform random symmetric matrix, no triangle inequality
        D = rng.random((n, n)) * 10.0
    elif kind == "negative":                      # symmetric with negative entries (operators.py:32-50,129-147 accept any matrix);
        # the mean stays positive: with a negative start-tour length k = 0.1 cost / n is negative, penalties make edges CHEAPER
        # and the reference's `while moves < perturbation_moves` (algorithms.py:150) never ends -- in the reference itself
        D = rng.normal(1.0, 0.6, size=(n, n))
    elif kind == "huge":                          # |D| > 1e6: the pruned scans' magnitude bound fails (prune_ok = 0: full scans run)
        pos = rng.random((n, 2)) * 1e7
        D = np.sqrt(((pos[:, None] - pos[None]) ** 2).sum(-1))
    else:                                         # lattice + noise around np.isclose's 1e-8 threshold
        pos = rng.integers(0, 4, size=(n, 2)).astype(np.float64)
        D = np.abs(pos[:, None] - pos[None]).sum(-1) + 1.0
        D = D + rng.choice([0.0, 5e-9, 1e-8, 1.00001e-8, 2e-8, -5e-9, -1e-8], size=D.shape)
    D = np.triu(D, 1)
    D = D + D.T
    g = np.maximum(rng.normal(0.0, 0.1, size=(n, n)).astype(np.float32).astype(np.float64), 0)   # ~half zeros
    g = np.triu(g, 1)
    return D, g + g.T


CASES = []
_rng = np.random.default_rng(987654)
for _ in range(48):
    CASES.append(dict(n=int(_rng.integers(4, 131)), kind=str(_rng.choice(["euclid", "lattice", "noisy"])),
                      pm=int(_rng.choice([1, 5, 20, 30])), fi=bool(_rng.integers(0, 2)), K=int(_rng.integers(1, 5)),
                      bits=int(_rng.choice([0, 16, 32])), guides=int(_rng.integers(1, 3)), seed=int(_rng.integers(1 << 30))))


# round 2: the 4-register-slot lean scans, 16-wave workgroups and the forced compact store (n = 128..255, bits -2)
_rng2 = np.random.default_rng(24680)
for _ in range(14):
    CASES.append(dict(n=int(_rng2.integers(120, 256)), kind=str(_rng2.choice(["euclid", "lattice", "noisy"])),
                      pm=int(_rng2.choice([5, 20])), fi=bool(_rng2.integers(0, 2)), K=int(_rng2.integers(1, 3)),
                      bits=int(_rng2.choice([0, -2])), guides=int(_rng2.integers(1, 3)), seed=int(_rng2.integers(1 << 30))))


# round 5: the pruned descent scans (n >= 80) and the edge form of the perturbation phase on matrices that are not
# distances -- non-metric, negative entries, magnitudes beyond the pruning argument's bound
_rng3 = np.random.default_rng(13579)
for _k in ["nonmetric", "negative", "huge"] * 4:
    CASES.append(dict(n=int(_rng3.integers(80, 256)), kind=_k, pm=int(_rng3.choice([5, 20])), fi=bool(_rng3.integers(0, 4) == 0),
                      K=int(_rng3.integers(1, 3)), bits=int(_rng3.choice([0, -2])), guides=int(_rng3.integers(1, 3)),
                      seed=int(_rng3.integers(1 << 30))))


# round 6: the quiet rows of the relocate scan (exact don't-look flags: 80 <= n <= 127 on every symmetric store, 128 <= n <= 163 on
# the LDS-penalty store) over several outer iterations, on matrices full of exact ties and of deltas around np.isclose's threshold
_rng4 = np.random.default_rng(97531)
for _k in ["lattice", "noisy", "euclid", "nonmetric"] * 3:
    CASES.append(dict(n=int(_rng4.integers(80, 164)), kind=_k, pm=int(_rng4.choice([5, 20])), fi=False,
                      K=int(_rng4.integers(4, 9)), bits=int(_rng4.choice([0, 0, -2])), guides=int(_rng4.integers(1, 3)),
                      seed=int(_rng4.integers(1 << 30))))


VARIANTS = {"serial": (0, 1), "team": (1, 1), "fullscan": (0, 0)}     # (team form of the perturbation phase, pruned descent scans)


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("c", CASES, ids=lambda c: f"n{c['n']}-{c['kind']}-pm{c['pm']}-fi{int(c['fi'])}-K{c['K']}-b{c['bits']}-g{c['guides']}")
def test_fuzz_case(c, variant):
    """Kernel variants, all bit-exact: `serial` = perturbation phase on wavefront 0 (what a device-filling batch runs);
    `team` = on all wavefronts of the workgroup wherever that form exists; both with the pruned descent scans where those
    exist (n >= 80, the default); `fullscan` = the descent evaluates every move of its scans."""
    team, prune = VARIANTS[variant]
    if not prune and c["n"] < 80:
        pytest.skip("the pruned descent scans only exist for n >= 80: same kernel as the serial case")
    from gnngls_amd import ops
    from oracle import gls_oracle as go
    if team and c["bits"] == 16:
        pytest.skip("the team form exists for 32-bit counters only: same kernel as the serial case")
    rng = np.random.default_rng(c["seed"])
    n, B = c["n"], 3
    Ds, Gs = zip(*[make_case(rng, n, c["kind"]) for _ in range(B)])
    D = np.stack(Ds)
    guides = np.stack([np.stack(Gs), D][:c["guides"]])            # [G,B,n,n]: regret-like guide, then 'weight'
    d = torch.from_numpy(D).cuda()
    gd = torch.from_numpy(np.ascontiguousarray(guides)).cuda()
    init = ops.nearest_neighbor(gd[0].contiguous())
    cost = ops.tour_cost(init, d)
    with ops.gls_team_mode(team), ops.gls_prune_mode(prune):
        cfg = ops.gls_describe_config(n, B, c["bits"], first_improvement=c["fi"])
        assert cfg["team"] == (bool(team) and cfg["store"] != "global")          # n > ~200: the triangles leave the LDS
        r = ops.gls_run(d, gd, init, cost, perturbation_moves=c["pm"], first_improvement=c["fi"], max_outer_iters=c["K"],
                        trace_cap=1 << 13, want_penalty=True, penalty_bits=c["bits"])
        if team:                                                 # and the trace-free instantiation of the same form
            plain = ops.gls_run(d, gd, init, cost, perturbation_moves=c["pm"], first_improvement=c["fi"],
                                max_outer_iters=c["K"], penalty_bits=c["bits"])
            assert torch.equal(plain.best_tour, r.best_tour) and torch.equal(plain.best_cost, r.best_cost)
            assert torch.equal(plain.trace_len, r.trace_len) and torch.equal(plain.evals, r.evals)
    init_h, cost_h = init.cpu().numpy(), cost.cpu().numpy()
    for b in range(B):
        assert init_h[b].tolist() == go.nearest_neighbor(guides[0, b])
        o = go.guided_local_search(D[b], guides[:, b], init_h[b], cost_h[b], perturbation_moves=c["pm"],
                                   first_improvement=c["fi"], max_outer_iters=c["K"])
        L = o["trace_len"]
        assert int(r.status[b]) == 0 and int(r.trace_len[b]) == L
        got = r.trace_cost[b, :L].cpu().numpy()
        assert np.array_equal(got.view(np.uint64), o["trace"].view(np.uint64)), \
            f"first mismatch at move {int(np.argmax(got.view(np.uint64) != o['trace'].view(np.uint64)))} of {L}"
        assert r.best_tour[b].cpu().tolist() == o["best_tour"]
        assert np.float64(r.best_cost[b].item()).tobytes() == np.float64(o["best_cost"]).tobytes()
        assert np.array_equal(r.penalty[b].cpu().numpy(), o["penalty"])


@pytest.mark.parametrize("n", [5, 17, 64, 101])
def test_operators_accept_asymmetric_matrices(n):
    """The operator-level entry points keep the reference's exact index order (D[a,c], D[b,d], D[a,b], D[c,d] ...),
    so an asymmetric matrix gives the reference's result too (the LDS-resident search kernel assumes symmetry)."""
    from gnngls_amd import ops
    from oracle import gls_oracle as go
    rng = np.random.default_rng(n)
    D = rng.random((n, n))
    np.fill_diagonal(D, 0.0)
    assert not np.array_equal(D, D.T)
    tour = np.concatenate([[0], 1 + rng.permutation(n - 1), [0]]).astype(np.int32)
    t = torch.from_numpy(tour[None]).cuda()
    d = torch.from_numpy(D[None]).cuda()
    a, b = ops.two_opt_delta_all(t, d)[0].cpu().numpy(), go.two_opt_delta_all(tour, D)
    m = ~np.isnan(b)
    assert np.array_equal(a[m].view(np.uint64), b[m].view(np.uint64))
    a, b = ops.relocate_delta_all(t, d)[0].cpu().numpy(), go.relocate_delta_all(tour, D)
    assert np.array_equal(a[m].view(np.uint64), b[m].view(np.uint64))
    for op, name in ((0, "two_opt"), (1, "relocate")):
        for fi in (False, True):
            delta, move, nt = ops.best_move(t, d, op, None, fi)
            od, ot, _ = getattr(go, name + "_a2a")(tour, D, fi)
            assert np.float64(delta.item()).tobytes() == np.float64(od).tobytes() and nt[0].cpu().tolist() == ot
            pos = torch.tensor([n // 2], dtype=torch.int32, device="cuda")
            delta, move, nt = ops.best_move(t, d, op, pos, fi)
            od, ot, _ = getattr(go, name + "_o2a")(tour, D, n // 2, fi)
            assert np.float64(delta.item()).tobytes() == np.float64(od).tobytes() and nt[0].cpu().tolist() == ot
    assert np.float64(ops.tour_cost(t, d).item()).tobytes() == np.float64(go.tour_cost(tour, D)).tobytes()


def test_local_search_mirror_rejects_asymmetric():
    from gnngls_amd import algorithms as alg
    D = np.random.default_rng(0).random((6, 6))
    with pytest.raises(NotImplementedError):
        alg.local_search([0, 1, 2, 3, 4, 5, 0], 1.0, D)


@pytest.mark.parametrize("n,bits", [(12, 0), (50, 0), (100, -2), (150, 0), (200, -2)])
def test_asymmetric_matrix_is_flagged_never_searched_silently(n, bits):
    """The symmetric stores keep D[max, min] only.  An instance whose matrix differs from its transpose by one ulp in ONE entry
    comes back with GNNGLS_STATUS_ASYMMETRIC, untouched (C ABI: never a silent different search); ops.gls_run reruns it on the
    global-memory store, whose result is what that store gives for the whole batch (the reference's index order)."""
    from gnngls_amd import ops
    rng = np.random.default_rng(n)
    B = 3
    Ds, Gs = zip(*[make_case(rng, n, "euclid") for _ in range(B)])
    D = np.stack(Ds)
    D[1, 2, 1] = np.nextafter(D[1, 2, 1], np.inf)                  # instance 1: D[2,1] != D[1,2] by one ulp
    guides = np.stack([np.stack(Gs)])
    d, gd = torch.from_numpy(D).cuda(), torch.from_numpy(np.ascontiguousarray(guides)).cuda()
    init = ops.nearest_neighbor(gd[0].contiguous())
    cost = ops.tour_cost(init, d)
    assert ops.gls_describe_config(n, B, bits)["store"] != "global"
    raw = ops.gls_run(d, gd, init, cost, max_outer_iters=2, penalty_bits=bits, retry_asymmetric=False, want_penalty=True, imp_cap=8)
    assert raw.status.cpu().tolist() == [0, ops.STATUS_ASYMMETRIC, 0]
    assert torch.equal(raw.best_tour[1], init[1]) and raw.best_cost[1].item() == cost[1].item()
    assert int(raw.outer_iters[1]) == 0 and int(raw.evals[1]) == 0 and int(raw.penalty[1].abs().sum()) == 0 and int(raw.imp_len[1]) == 0
    auto = ops.gls_run(d, gd, init, cost, max_outer_iters=2, penalty_bits=bits, want_penalty=True)
    glob = ops.gls_run(d, gd, init, cost, max_outer_iters=2, penalty_bits=-1, want_penalty=True)
    assert auto.status.cpu().tolist() == [0, 0, 0] and glob.status.cpu().tolist() == [0, 0, 0]
    for name in ("best_tour", "best_cost", "outer_iters", "evals", "penalty"):
        assert torch.equal(getattr(auto, name), getattr(glob, name)), name
    assert int(auto.outer_iters[1]) == 2
