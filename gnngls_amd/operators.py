"""Mirror of gnngls/operators.py (reference operators.py:6-147): same names, argument meaning,
return values and error behaviour, evaluated by the HIP kernels through the C ABI.

`tour` is a Python list of length n+1 with the depot at both ends, `D` an n x n array-like indexed by
node label.  a2a / o2a return `(delta, new_tour)` and never mutate their inputs.  These wrappers move
one tour per call to the GPU -- they exist for drop-in compatibility and for parity tests; the fast
path is gnngls_amd.ops / gnngls_amd.pipeline (whole batches, no host round trip per move).
"""
import numpy as np
import torch

from . import ops


def _dev(tour, D):
    t = ops.as_dev(np.asarray(tour, dtype=np.int32)[None], torch.int32)
    d = ops.as_dev(np.asarray(D, dtype=np.float64)[None], torch.float64)
    return t, d


def two_opt(tour, i, j):
    """operators.py:6-11"""
    if i == j:
        return tour
    elif j < i:
        i, j = j, i
    return tour[:i] + tour[j - 1:i - 1:-1] + tour[j:]


def two_opt_cost(tour, D, i, j):
    """operators.py:14-29"""
    if i == j:
        return 0
    t, d = _dev(tour, D)
    return ops.two_opt_delta_all(t, d)[0, i, j].item()


def _scan(op, tour, D, i, first_improvement, apply):
    t, d = _dev(tour, D)
    pos = None if i is None else torch.tensor([i], dtype=torch.int32, device=t.device)
    delta, move, _ = ops.best_move(t, d, op, pos, first_improvement)
    mi, mj = move[0].tolist()
    if mi == 0 and mj == 0:
        return 0, tour
    return delta.item(), apply(tour, mi, mj)


def two_opt_a2a(tour, D, first_improvement=False):
    """operators.py:32-50"""
    return _scan(ops.OP_TWO_OPT, tour, D, None, first_improvement, two_opt)


def two_opt_o2a(tour, D, i, first_improvement=False):
    """operators.py:53-73"""
    assert i > 0 and i < len(tour) - 1
    return _scan(ops.OP_TWO_OPT, tour, D, i, first_improvement, two_opt)


def relocate(tour, i, j):
    """operators.py:76-80"""
    new_tour = tour.copy()
    n = new_tour.pop(i)
    new_tour.insert(j, n)
    return new_tour


def relocate_cost(tour, D, i, j):
    """operators.py:83-103"""
    if i == j:
        return 0
    t, d = _dev(tour, D)
    return ops.relocate_delta_all(t, d)[0, i, j].item()


def relocate_o2a(tour, D, i, first_improvement=False):
    """operators.py:106-126"""
    assert i > 0 and i < len(tour) - 1
    return _scan(ops.OP_RELOCATE, tour, D, i, first_improvement, relocate)


def relocate_a2a(tour, D, first_improvement=False):
    """operators.py:129-147"""
    return _scan(ops.OP_RELOCATE, tour, D, None, first_improvement, relocate)
