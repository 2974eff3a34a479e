"""oracle/bnb_tsp.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/bnb_tsp.c: exact TSP optima by branch and bound on the Held-Karp 1-tree bound -- the denominator
the reference's gap uses (scripts/test.py:62,104 divides by the Concorde optimum of its instance files, which are git-LFS
stubs here).  Only tests/, scripts/make_exact_optima.py and bench.py's reporting (outside the timed region) use it."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbnb_tsp.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "bnb_tsp.c")
        if not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libbnb_tsp.so"], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(_SO)
        L.bnb_tsp.restype = ctypes.c_double
        L.bnb_tsp.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_double, ctypes.c_long,
                              ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_int),
                              ctypes.POINTER(ctypes.c_int)]
        _lib = L
    return _lib


def solve(D, ub, max_nodes=200000):
    """D [n,n] symmetric fp64, ub = length of a known tour -> dict(value, proven, nodes, tour or None): value is the optimum
    when proven (min(ub, shortest tour the search found)), else the best upper bound."""
    D = np.ascontiguousarray(D, dtype=np.float64)
    n = D.shape[0]
    proven, found, nodes = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_long(0)
    tour = (ctypes.c_int * n)()
    v = lib().bnb_tsp(D.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n, float(ub), int(max_nodes), ctypes.byref(proven),
                      ctypes.byref(nodes), tour, ctypes.byref(found))
    return {"value": float(v), "proven": bool(proven.value), "nodes": int(nodes.value), "tour": list(tour) if found.value else None}
