"""Mirror of the hot-path functions of gnngls/algorithms.py (reference algorithms.py:9-18,111-195):
`nearest_neighbor`, `local_search`, `guided_local_search` with the reference's signatures, return
values and side effects, executed by the persistent HIP search kernel.

The alternative tour constructors of the reference (algorithms.py:21-108) are never called by any
script and are out of scope (SURVEY.md C3').
"""
import time
import warnings

import networkx as nx
import numpy as np
import torch

from . import ops

TRACE_CAP = 1 << 16          # per-move trace entries that are always available
TRACE_PER_SECOND = 500_000   # extra entries per second of budget (a TSP100 search accepts ~2e5 moves per second)
TRACE_CAP_MAX = 1 << 25      # 400 MB of device memory for one instance's trace: beyond this the record degrades (see _progress)
IMP_CAP = 4096


class SearchAborted(RuntimeWarning):
    """The device watchdog stopped a search early: the returned tour is the best found so far."""


def _trace_cap(budget_s):
    return int(min(TRACE_CAP + TRACE_PER_SECOND * max(budget_s, 0.0), TRACE_CAP_MAX))


def _attr_matrix(G, attr):
    """nx.attr_matrix(G, attr)[0] for nodes 0..n-1 (algorithms.py:140,163): symmetric n x n fp64."""
    n = len(G.nodes)
    M = np.zeros((n, n), dtype=np.float64)
    for u, v, w in G.edges(data=attr):
        M[u, v] = w
        M[v, u] = w
    return M


def nearest_neighbor(G, depot, weight="weight"):
    """algorithms.py:9-18 (greedy on G.edges[(i,j)][weight]; ties -> first neighbour = lowest id)."""
    W = ops.as_dev(_attr_matrix(G, weight)[None], torch.float64)
    return ops.nearest_neighbor(W, depot)[0].tolist()


def _check_status(r, what):
    st = r.status.cpu().numpy()
    if (st == ops.STATUS_WATCHDOG).any():
        warnings.warn(f"{what}: the device watchdog stopped {int((st == ops.STATUS_WATCHDOG).sum())} search(es) early; "
                      "the result is the best tour found so far (pass a larger watchdog_s)", SearchAborted, stacklevel=3)


def _progress(r, t_host):
    """search_progress of the reference: one {'time','cost'} per accepted move (algorithms.py:127-130,180-183).
    If more moves were accepted than the device trace holds, the tail is the bounded improvement record (the returned
    best whenever it improved) and a warning says so -- the list never silently stops short of the returned cost."""
    L, cap = int(r.trace_len[0]), r.trace_cost.shape[1]
    kept = min(L, cap)
    costs = r.trace_cost[0, :kept].tolist()
    times = r.trace_time[0, :kept].tolist()
    rows = [{"time": t_host + dt, "cost": c} for dt, c in zip(times, costs)]
    if L > cap:
        warnings.warn(f"search trace truncated: {L} accepted moves, {cap} kept; later entries are new-best events only",
                      RuntimeWarning, stacklevel=3)
        k = min(int(r.imp_len[0]), r.imp_cost.shape[1])
        t_cut = times[-1] if times else -1.0
        rows += [{"time": t_host + dt, "cost": c}
                 for dt, c in zip(r.imp_time[0, :k].tolist(), r.imp_cost[0, :k].tolist()) if dt > t_cut]
    return rows


def local_search(init_tour, init_cost, D, first_improvement=False):
    """algorithms.py:111-132 -> (cur_tour, cur_cost, search_progress)."""
    D = np.asarray(D, dtype=np.float64)
    if not np.array_equal(D, D.T):
        # two_opt_cost assumes D[x,y] == D[y,x] (a reversal flips every inner edge), so on an asymmetric matrix the
        # reference's descent is not a descent at all and need not terminate; refuse instead of spinning
        raise NotImplementedError("local_search needs a symmetric distance matrix "
                                  "(nx.attr_matrix of an undirected graph always is)")
    t0 = time.time()
    r = ops.gls_run(ops.as_dev(D[None], torch.float64), None,
                    ops.as_dev(np.asarray(init_tour, dtype=np.int32)[None], torch.int32),
                    ops.as_dev(np.asarray([init_cost], dtype=np.float64), torch.float64),
                    first_improvement=first_improvement, max_outer_iters=0, trace_cap=TRACE_CAP, want_trace_time=True,
                    imp_cap=IMP_CAP)
    _check_status(r, "local_search")
    return r.best_tour[0].tolist(), r.best_cost[0].item(), _progress(r, t0)


def guided_local_search(G, init_tour, init_cost, t_lim, weight="weight", guides=["weight"], perturbation_moves=30,
                        first_improvement=False, max_outer_iters=None, watchdog_s=None):
    """algorithms.py:135-195 -> (best_tour, best_cost, search_progress).

    `t_lim` is an absolute time.time() deadline as in the reference.  Like the reference this
    writes the final 'penalty' edge attribute into G (algorithms.py:138,161).  `max_outer_iters`
    (extension) runs an exact number of outer iterations instead of the wall-clock budget; `watchdog_s` (extension)
    bounds the device time of the run (default: budget + 5 s, or a bound scaled to the requested iterations) -- a run
    stopped by it returns the best tour so far and raises a SearchAborted warning."""
    D = _attr_matrix(G, weight)                                                  # algorithms.py:140
    gm = np.stack([_attr_matrix(G, g) for g in guides])[:, None]                 # [G,1,n,n]
    t0 = time.time()
    remaining = max(t_lim - t0, 0.0)
    r = ops.gls_run(ops.as_dev(D[None], torch.float64), ops.as_dev(gm, torch.float64),
                    ops.as_dev(np.asarray(init_tour, dtype=np.int32)[None], torch.int32),
                    ops.as_dev(np.asarray([init_cost], dtype=np.float64), torch.float64),
                    perturbation_moves=perturbation_moves, first_improvement=first_improvement,
                    max_outer_iters=-1 if max_outer_iters is None else int(max_outer_iters), time_limit_s=remaining,
                    watchdog_s=watchdog_s, trace_cap=_trace_cap(remaining if max_outer_iters is None else 0.0),
                    want_trace_time=True, want_penalty=True, imp_cap=IMP_CAP)
    _check_status(r, "guided_local_search")
    pen = r.penalty[0].cpu().numpy()
    nx.set_edge_attributes(G, {(u, v): float(pen[u, v]) for u, v in G.edges}, "penalty")
    return r.best_tour[0].tolist(), r.best_cost[0].item(), _progress(r, t0)
