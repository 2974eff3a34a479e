"""oracle/gls_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/gls_oracle.c (the scalar CPU restatement of gnngls/operators.py and
gnngls/algorithms.py:9-18,111-195).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this.  Parity: pinned against tests/golden/*.npz (see gls_oracle.c header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GLS_ORACLE_SO: an alternative build of the same source (tests/test_sanitizers_cpu.py loads an ASan/UBSan build)
_SO = os.environ.get("GLS_ORACLE_SO") or os.path.join(_HERE, "libgls_oracle.so")
_lib = None

_f64p = ctypes.POINTER(ctypes.c_double)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_intp = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    src = os.path.join(_HERE, "gls_oracle.c")
    if force or not os.path.isfile(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgls_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.gls_oracle_two_opt_cost.restype = ctypes.c_double
        L.gls_oracle_two_opt_cost.argtypes = [_i32p, _f64p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.gls_oracle_relocate_cost.restype = ctypes.c_double
        L.gls_oracle_relocate_cost.argtypes = [_i32p, _f64p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.gls_oracle_two_opt.argtypes = [_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.gls_oracle_relocate.argtypes = [_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        for name in ("gls_oracle_two_opt_a2a", "gls_oracle_relocate_a2a"):
            f = getattr(L, name)
            f.restype = ctypes.c_int
            f.argtypes = [_i32p, _f64p, ctypes.c_int, ctypes.c_int, _f64p, _intp, _intp]
        for name in ("gls_oracle_two_opt_o2a", "gls_oracle_relocate_o2a"):
            f = getattr(L, name)
            f.restype = ctypes.c_int
            f.argtypes = [_i32p, _f64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _f64p, _intp]
        L.gls_oracle_tour_cost.restype = ctypes.c_double
        L.gls_oracle_tour_cost.argtypes = [_i32p, _f64p, ctypes.c_int]
        L.gls_oracle_nearest_neighbor.argtypes = [_f64p, ctypes.c_int, ctypes.c_int, _i32p]
        L.gls_oracle_local_search.restype = ctypes.c_int
        L.gls_oracle_local_search.argtypes = [_i32p, _f64p, _f64p, ctypes.c_int, ctypes.c_int,
                                              _f64p, ctypes.c_int, _intp]
        L.gls_oracle_guided_local_search.restype = ctypes.c_double
        L.gls_oracle_guided_local_search.argtypes = [
            _f64p, _f64p, ctypes.c_int, ctypes.c_int, _i32p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
            ctypes.c_int64, ctypes.c_double, _f64p, ctypes.c_int, _intp, _i32p, _i64p, _i64p,
            _f64p, _i64p, ctypes.c_int, _intp]
        L.gls_oracle_guided_local_search_timed.restype = ctypes.c_double
        L.gls_oracle_guided_local_search_timed.argtypes = L.gls_oracle_guided_local_search.argtypes + [_f64p]
        for name in ("gls_oracle_two_opt_delta_all", "gls_oracle_relocate_delta_all"):
            getattr(L, name).argtypes = [_i32p, _f64p, ctypes.c_int, _f64p]
        _lib = L
    return _lib


def _t(tour):
    return np.ascontiguousarray(np.asarray(tour, dtype=np.int32))


def _d(D):
    return np.ascontiguousarray(np.asarray(D, dtype=np.float64))


def _p(a, ty):
    return a.ctypes.data_as(ty)


def two_opt_cost(tour, D, i, j):
    t, D = _t(tour), _d(D)
    return lib().gls_oracle_two_opt_cost(_p(t, _i32p), _p(D, _f64p), len(t) - 1, i, j)


def relocate_cost(tour, D, i, j):
    t, D = _t(tour), _d(D)
    return lib().gls_oracle_relocate_cost(_p(t, _i32p), _p(D, _f64p), len(t) - 1, i, j)


def two_opt(tour, i, j):
    t = _t(tour).copy()
    lib().gls_oracle_two_opt(_p(t, _i32p), len(t) - 1, i, j)
    return t.tolist()


def relocate(tour, i, j):
    t = _t(tour).copy()
    lib().gls_oracle_relocate(_p(t, _i32p), len(t) - 1, i, j)
    return t.tolist()


def _a2a(name, apply, tour, D, first_improvement):
    t, D = _t(tour), _d(D)
    delta = ctypes.c_double(0.0)
    bi, bj = ctypes.c_int(0), ctypes.c_int(0)
    found = getattr(lib(), name)(_p(t, _i32p), _p(D, _f64p), len(t) - 1, int(first_improvement),
                                 ctypes.byref(delta), ctypes.byref(bi), ctypes.byref(bj))
    if found:
        return delta.value, apply(t, bi.value, bj.value), (bi.value, bj.value)
    return 0, t.tolist(), None


def _o2a(name, apply, tour, D, i, first_improvement):
    t, D = _t(tour), _d(D)
    assert i > 0 and i < len(t) - 1
    delta = ctypes.c_double(0.0)
    bj = ctypes.c_int(0)
    found = getattr(lib(), name)(_p(t, _i32p), _p(D, _f64p), len(t) - 1, i, int(first_improvement),
                                 ctypes.byref(delta), ctypes.byref(bj))
    if found:
        return delta.value, apply(t, i, bj.value), (i, bj.value)
    return 0, t.tolist(), None


def two_opt_a2a(tour, D, first_improvement=False):
    return _a2a("gls_oracle_two_opt_a2a", two_opt, tour, D, first_improvement)


def relocate_a2a(tour, D, first_improvement=False):
    return _a2a("gls_oracle_relocate_a2a", relocate, tour, D, first_improvement)


def two_opt_o2a(tour, D, i, first_improvement=False):
    return _o2a("gls_oracle_two_opt_o2a", two_opt, tour, D, i, first_improvement)


def relocate_o2a(tour, D, i, first_improvement=False):
    return _o2a("gls_oracle_relocate_o2a", relocate, tour, D, i, first_improvement)


def tour_cost(tour, D):
    t, D = _t(tour), _d(D)
    return lib().gls_oracle_tour_cost(_p(t, _i32p), _p(D, _f64p), len(t) - 1)


def nearest_neighbor(W, depot=0):
    W = _d(W)
    n = W.shape[0]
    t = np.zeros(n + 1, dtype=np.int32)
    lib().gls_oracle_nearest_neighbor(_p(W, _f64p), n, depot, _p(t, _i32p))
    return t.tolist()


def local_search(init_tour, init_cost, D, first_improvement=False, trace_cap=1 << 16):
    t, D = _t(init_tour).copy(), _d(D)
    cost = ctypes.c_double(float(init_cost))
    trace = np.zeros(trace_cap, dtype=np.float64)
    tl = ctypes.c_int(0)
    lib().gls_oracle_local_search(_p(t, _i32p), ctypes.byref(cost), _p(D, _f64p), len(t) - 1,
                                  int(first_improvement), _p(trace, _f64p), trace_cap, ctypes.byref(tl))
    return t.tolist(), cost.value, trace[:min(tl.value, trace_cap)].copy()


def guided_local_search(D, guides, init_tour, init_cost, perturbation_moves=30, first_improvement=False,
                        max_outer_iters=-1, time_limit_s=0.0, trace_cap=1 << 20, want_penalty=True, imp_cap=4096):
    """guides: array [G, n, n] fp64.  Returns dict(best_tour, best_cost, trace, penalty, outer_iters, evals,
    imp_cost, imp_iter, imp_time, imp_len) -- imp_* = the returned best at every improvement (algorithms.py:143,190-191),
    imp_time = seconds of wall clock since the search started."""
    D = _d(D)
    guides = _d(guides)
    if guides.ndim == 2:
        guides = guides[None]
    n = D.shape[0]
    t = _t(init_tour).copy()
    trace = np.zeros(max(trace_cap, 1), dtype=np.float64)
    tl = ctypes.c_int(0)
    pen = np.zeros((n, n), dtype=np.int32)
    iters = ctypes.c_int64(0)
    evals = ctypes.c_int64(0)
    imp_cost = np.zeros(max(imp_cap, 1), dtype=np.float64)
    imp_iter = np.zeros(max(imp_cap, 1), dtype=np.int64)
    imp_time = np.zeros(max(imp_cap, 1), dtype=np.float64)
    il = ctypes.c_int(0)
    best = lib().gls_oracle_guided_local_search_timed(
        _p(D, _f64p), _p(guides, _f64p), guides.shape[0], n, _p(t, _i32p), float(init_cost),
        int(perturbation_moves), int(first_improvement), int(max_outer_iters), float(time_limit_s),
        _p(trace, _f64p), trace_cap, ctypes.byref(tl), _p(pen, _i32p) if want_penalty else None,
        ctypes.byref(iters), ctypes.byref(evals), _p(imp_cost, _f64p), _p(imp_iter, _i64p), imp_cap, ctypes.byref(il),
        _p(imp_time, _f64p))
    k = min(il.value, imp_cap)
    return dict(best_tour=t.tolist(), best_cost=best, trace=trace[:min(tl.value, trace_cap)].copy(),
                trace_len=tl.value, penalty=pen, outer_iters=iters.value, evals=evals.value,
                imp_cost=imp_cost[:k].copy(), imp_iter=imp_iter[:k].copy(), imp_time=imp_time[:k].copy(), imp_len=il.value)


def two_opt_delta_all(tour, D):
    t, D = _t(tour), _d(D)
    n = len(t) - 1
    out = np.zeros((n + 1, n + 1), dtype=np.float64)
    lib().gls_oracle_two_opt_delta_all(_p(t, _i32p), _p(D, _f64p), n, _p(out, _f64p))
    return out


def relocate_delta_all(tour, D):
    t, D = _t(tour), _d(D)
    n = len(t) - 1
    out = np.zeros((n + 1, n + 1), dtype=np.float64)
    lib().gls_oracle_relocate_delta_all(_p(t, _i32p), _p(D, _f64p), n, _p(out, _f64p))
    return out
