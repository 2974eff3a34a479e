"""oracle/one_tree.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/one_tree.c: Held-Karp 1-tree lower bounds of TSP optima (the far side of the optimality-gap
bracket bench.py reports above the reach of the exact DP; the reference divides by Concorde's optimum, scripts/test.py:62,104).
Only tests/ and bench.py's reporting (outside the timed region) import this."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("ONE_TREE_SO") or os.path.join(_HERE, "libone_tree.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "one_tree.c")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libone_tree.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if "ONE_TREE_SO" not in os.environ:
            build()
        L = ctypes.CDLL(_SO)
        L.one_tree_lower_bound.restype = ctypes.c_double
        L.one_tree_lower_bound.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_double, ctypes.c_int]
        _lib = L
    return _lib


def lower_bound(D, ub, max_iters=2000):
    """D [n,n] symmetric fp64, ub = length of any tour -> Held-Karp lower bound of the optimal tour length."""
    D = np.ascontiguousarray(D, dtype=np.float64)
    return float(lib().one_tree_lower_bound(D.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), D.shape[0], float(ub),
                                            int(max_iters)))


def _one(args):
    return lower_bound(*args)


def lower_bounds(Ds, ubs, workers=None, max_iters=2000):
    """Bounds of a batch [B,n,n] given per-instance upper bounds (tour lengths), one host process per worker."""
    Ds = np.ascontiguousarray(Ds, dtype=np.float64)
    ubs = np.asarray(ubs, dtype=np.float64)
    workers = workers or min(len(Ds), os.cpu_count() or 1, 32)
    jobs = [(D, float(u), max_iters) for D, u in zip(Ds, ubs)]
    if workers <= 1:
        return np.array([_one(j) for j in jobs])
    import multiprocessing as mp
    build()
    with mp.get_context("spawn").Pool(workers) as pool:
        return np.array(pool.map(_one, jobs, chunksize=max(1, len(jobs) // (4 * workers))))
