"""oracle/held_karp.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/held_karp.c: exact TSP optimum for n <= 21 (the gap denominator of SURVEY.md 8(d) config 1 in
place of the Concorde labels the reference reads from its LFS-stored instance files, scripts/test.py:62,104)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("HELD_KARP_SO") or os.path.join(_HERE, "libheld_karp.so")   # alternative (sanitizer) build
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "held_karp.c")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libheld_karp.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if "HELD_KARP_SO" not in os.environ:
            build()
        L = ctypes.CDLL(_SO)
        L.held_karp.restype = ctypes.c_double
        L.held_karp.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_int32)]
        _lib = L
    return _lib


def optimum(D):
    """D [n,n] fp64 -> (optimal tour cost summed like gnngls.tour_cost, optimal tour list[n+1] from depot 0)."""
    D = np.ascontiguousarray(D, dtype=np.float64)
    n = D.shape[0]
    tour = np.zeros(n + 1, dtype=np.int32)
    cost = lib().held_karp(D.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n,
                           tour.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if cost < 0:
        raise ValueError(f"held_karp: n={n} out of range (2..21) or out of memory")
    return float(cost), tour.tolist()


def _one(D):
    return optimum(D)[0]


def optima(Ds, workers=None):
    """Optima of a batch [B,n,n], one host process per worker (the table of one TSP20 instance is 80 MB)."""
    Ds = np.ascontiguousarray(Ds, dtype=np.float64)
    workers = workers or min(len(Ds), os.cpu_count() or 1, 32)
    if workers <= 1:
        return np.array([_one(D) for D in Ds])
    import multiprocessing as mp
    build()
    with mp.get_context("spawn").Pool(workers) as pool:
        return np.array(pool.map(_one, list(Ds), chunksize=max(1, len(Ds) // (4 * workers))))
