"""Batched device API over the C ABI (include/gnngls_hip.h).

All tensors are torch CUDA tensors used as plain device buffers; every function enqueues HIP
kernels on the current torch stream through libgnngls_hip.so.  B = instances, n = nodes.
The reference-signature mirrors (operators.py / algorithms.py of this package) are thin
list-in/list-out wrappers over these.
"""
import ctypes
from dataclasses import dataclass

import torch

from . import _lib

OP_TWO_OPT = 0
OP_RELOCATE = 1


def _dev():
    if not torch.cuda.is_available():
        raise _lib.GnnglsHipError("no HIP device visible: gnngls_amd has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def as_dev(x, dtype):
    t = torch.as_tensor(x, dtype=dtype)
    return t.to(_dev()).contiguous()


def _check_shapes(tour, D):
    B, n1 = tour.shape
    n = n1 - 1
    assert D.shape == (B, n, n), f"D shape {tuple(D.shape)} does not match tours [{B},{n1}]"
    assert tour.dtype == torch.int32 and D.dtype == torch.float64
    return B, n


def two_opt_delta_all(tour, D):
    """[B,n+1,n+1] fp64 table of two_opt_cost (reference operators.py:14-29)."""
    B, n = _check_shapes(tour, D)
    out = torch.empty((B, n + 1, n + 1), dtype=torch.float64, device=tour.device)
    L = _lib.load()
    _lib.check(L.gnngls_two_opt_delta_all(_lib.ptr(tour), _lib.ptr(D), B, n, _lib.ptr(out), _lib.current_stream()),
               "two_opt_delta_all")
    return out


def relocate_delta_all(tour, D):
    """[B,n+1,n+1] fp64 table of relocate_cost (reference operators.py:83-103)."""
    B, n = _check_shapes(tour, D)
    out = torch.empty((B, n + 1, n + 1), dtype=torch.float64, device=tour.device)
    L = _lib.load()
    _lib.check(L.gnngls_relocate_delta_all(_lib.ptr(tour), _lib.ptr(D), B, n, _lib.ptr(out), _lib.current_stream()),
               "relocate_delta_all")
    return out


def best_move(tour, D, op, pos_i=None, first_improvement=False):
    """One a2a (pos_i None) or o2a scan.  Returns (delta[B], move[B,2], new_tour[B,n+1])."""
    B, n = _check_shapes(tour, D)
    delta = torch.empty((B,), dtype=torch.float64, device=tour.device)
    move = torch.empty((B, 2), dtype=torch.int32, device=tour.device)
    new_tour = torch.empty_like(tour)
    if pos_i is not None:
        assert pos_i.dtype == torch.int32 and pos_i.shape == (B,)
    L = _lib.load()
    _lib.check(L.gnngls_best_move(_lib.ptr(tour), _lib.ptr(D), B, n, int(op), _lib.ptr(pos_i), int(first_improvement),
                                  _lib.ptr(delta), _lib.ptr(move), _lib.ptr(new_tour), _lib.current_stream()),
               "best_move")
    return delta, move, new_tour


def tour_cost(tour, D):
    B, n = _check_shapes(tour, D)
    out = torch.empty((B,), dtype=torch.float64, device=tour.device)
    L = _lib.load()
    _lib.check(L.gnngls_tour_cost(_lib.ptr(tour), _lib.ptr(D), B, n, _lib.ptr(out), _lib.current_stream()), "tour_cost")
    return out


def nearest_neighbor(W, depot=0):
    """W [B,n,n] fp64 -> tours [B,n+1] int32 (reference algorithms.py:9-18)."""
    B, n, n2 = W.shape
    assert n == n2 and W.dtype == torch.float64
    out = torch.empty((B, n + 1), dtype=torch.int32, device=W.device)
    L = _lib.load()
    _lib.check(L.gnngls_nearest_neighbor(_lib.ptr(W), B, n, int(depot), _lib.ptr(out), _lib.current_stream()),
               "nearest_neighbor")
    return out


@dataclass
class GlsResult:
    best_tour: torch.Tensor      # [B,n+1] int32
    best_cost: torch.Tensor      # [B] fp64
    outer_iters: torch.Tensor    # [B] int64
    trace_cost: torch.Tensor     # [B,T] fp64 or None
    trace_time: torch.Tensor     # [B,T] fp32 or None
    trace_len: torch.Tensor      # [B] int32
    penalty: torch.Tensor        # [B,n,n] int32 or None
    evals: torch.Tensor          # [B] int64
    status: torch.Tensor         # [B] int32
    imp_cost: torch.Tensor = None   # [B,imp_cap] fp64: the returned best after every improvement (algorithms.py:143,190-191)
    imp_time: torch.Tensor = None   # [B,imp_cap] fp32 seconds since the workgroup started
    imp_iter: torch.Tensor = None   # [B,imp_cap] int64 completed outer iterations at that moment
    imp_len: torch.Tensor = None    # [B] int32 number of improvements (may exceed imp_cap)

    @property
    def trace_truncated(self):
        """[B] bool: more moves were accepted than the per-move trace could hold."""
        cap = 0 if self.trace_cost is None else self.trace_cost.shape[1]
        return self.trace_len > cap


STATUS_OK, STATUS_WATCHDOG, STATUS_PENALTY_OVERFLOW, STATUS_ASYMMETRIC = 0, 1, 2, 3


def gls_run(D, guides, init_tour, init_cost, perturbation_moves=30, first_improvement=False,
            max_outer_iters=-1, time_limit_s=0.0, watchdog_s=None, trace_cap=0, want_trace_time=False,
            want_penalty=False, penalty_bits=0, retry_overflow=True, imp_cap=0, retry_asymmetric=True):
    """guided_local_search (reference algorithms.py:135-195) for a batch of instances.

    D [B,n,n] fp64 symmetric, guides [G,B,n,n] fp64 (or None when max_outer_iters == 0 ->
    local_search only), init_tour [B,n+1] int32, init_cost [B] fp64.
    penalty_bits: 0 = auto (see include/gnngls_hip.h).  With retry_overflow (default) instances whose
    16-bit penalty counters overflowed are rerun with 32-bit counters (needs a host sync).
    (Such a rerun gets the same time_limit_s again: with penalty_bits=16 a wall-clock run can take twice the limit.)
    With retry_asymmetric (default) instances whose matrix is not bitwise symmetric (status STATUS_ASYMMETRIC: flagged on the
    device, not searched) are rerun on the global-memory store (penalty_bits = -1), which accepts any matrix (needs a host sync).
    imp_cap > 0 records the improvement trace (bounded however long the run is; see include/gnngls_hip.h).
    watchdog_s: None = time_limit_s + 5 s in wall-clock mode; in iteration-count mode a bound that scales with the
    requested work (the caller asked for exactly that many iterations, so the watchdog only catches hangs)."""
    B, n = _check_shapes(init_tour, D)
    dev = D.device
    G = 0
    if guides is not None:
        assert guides.dtype == torch.float64 and guides.dim() == 4 and guides.shape[1:] == (B, n, n)
        G = guides.shape[0]
    assert init_cost.dtype == torch.float64 and init_cost.shape == (B,)
    if watchdog_s is None:
        # iteration-count mode: ~n^2 evaluations per descent pass, tens of passes per outer iteration
        watchdog_s = (time_limit_s + 5.0) if max_outer_iters < 0 else 60.0 + 1e-7 * n * n * max(max_outer_iters, 1)
    best_tour = torch.empty_like(init_tour)
    best_cost = torch.empty((B,), dtype=torch.float64, device=dev)
    outer = torch.zeros((B,), dtype=torch.int64, device=dev)
    trace_cost = torch.zeros((B, trace_cap), dtype=torch.float64, device=dev) if trace_cap > 0 else None
    trace_time = torch.zeros((B, trace_cap), dtype=torch.float32, device=dev) if (trace_cap > 0 and want_trace_time) else None
    trace_len = torch.zeros((B,), dtype=torch.int32, device=dev)
    penalty = torch.zeros((B, n, n), dtype=torch.int32, device=dev) if want_penalty else None
    evals = torch.zeros((B,), dtype=torch.int64, device=dev)
    status = torch.zeros((B,), dtype=torch.int32, device=dev)
    imp_cost = torch.zeros((B, imp_cap), dtype=torch.float64, device=dev) if imp_cap > 0 else None
    imp_time = torch.zeros((B, imp_cap), dtype=torch.float32, device=dev) if imp_cap > 0 else None
    imp_iter = torch.zeros((B, imp_cap), dtype=torch.int64, device=dev) if imp_cap > 0 else None
    imp_len = torch.zeros((B,), dtype=torch.int32, device=dev) if imp_cap > 0 else None
    L = _lib.load()
    _lib.check(L.gnngls_gls_run(
        _lib.ptr(D), _lib.ptr(guides), G, B, n, _lib.ptr(init_tour), _lib.ptr(init_cost),
        int(perturbation_moves), int(first_improvement), int(penalty_bits), ctypes.c_int64(int(max_outer_iters)),
        float(time_limit_s), float(watchdog_s),
        _lib.ptr(best_tour), _lib.ptr(best_cost), _lib.ptr(outer),
        _lib.ptr(trace_cost), _lib.ptr(trace_time), int(trace_cap), _lib.ptr(trace_len),
        _lib.ptr(penalty), _lib.ptr(evals), _lib.ptr(status),
        _lib.ptr(imp_cost), _lib.ptr(imp_time), _lib.ptr(imp_iter), int(imp_cap), _lib.ptr(imp_len),
        _lib.current_stream()), "gls_run")
    res = GlsResult(best_tour, best_cost, outer, trace_cost, trace_time, trace_len, penalty, evals, status,
                    imp_cost, imp_time, imp_iter, imp_len)
    if retry_asymmetric and penalty_bits != -1:
        # instances whose matrix is not bitwise symmetric were flagged, not searched (the symmetric stores keep one triangle):
        # rerun them on the global-memory store, which follows the reference's index order (operators.py:25-28,97-102)
        bad = (status == STATUS_ASYMMETRIC).nonzero().flatten()
        if bad.numel() > 0:
            sub = gls_run(D[bad].contiguous(), None if guides is None else guides[:, bad].contiguous(),
                          init_tour[bad].contiguous(), init_cost[bad].contiguous(), perturbation_moves,
                          first_improvement, max_outer_iters, time_limit_s, watchdog_s, trace_cap, want_trace_time,
                          want_penalty, penalty_bits=-1, retry_overflow=False, imp_cap=imp_cap, retry_asymmetric=False)
            for name in ("best_tour", "best_cost", "outer_iters", "trace_cost", "trace_time", "trace_len", "penalty",
                         "evals", "status", "imp_cost", "imp_time", "imp_iter", "imp_len"):
                dst, src = getattr(res, name), getattr(sub, name)
                if dst is not None:
                    dst[bad] = src
    if retry_overflow and penalty_bits in (0, 16):
        bad = (status == STATUS_PENALTY_OVERFLOW).nonzero().flatten()
        if bad.numel() > 0:
            sub = gls_run(D[bad].contiguous(), None if guides is None else guides[:, bad].contiguous(),
                          init_tour[bad].contiguous(), init_cost[bad].contiguous(), perturbation_moves,
                          first_improvement, max_outer_iters, time_limit_s, watchdog_s, trace_cap, want_trace_time,
                          want_penalty, penalty_bits=32, retry_overflow=False, imp_cap=imp_cap)
            for name in ("best_tour", "best_cost", "outer_iters", "trace_cost", "trace_time", "trace_len", "penalty",
                         "evals", "status", "imp_cost", "imp_time", "imp_iter", "imp_len"):
                dst, src = getattr(res, name), getattr(sub, name)
                if dst is not None:
                    dst[bad] = src
    return res


def gls_kernel_resources(n, B=0, penalty_bits=0, first_improvement=False, trace=False):
    """-> dict(vgprs, scratch_bytes) of the kernel instantiation gls_run would launch (needs the device)."""
    v, sc = ctypes.c_int(0), ctypes.c_int(0)
    _lib.check(_lib.load().gnngls_gls_kernel_resources(int(n), int(B), int(penalty_bits), int(bool(first_improvement)), int(bool(trace)),
                                                       ctypes.cast(ctypes.byref(v), ctypes.c_void_p), ctypes.cast(ctypes.byref(sc), ctypes.c_void_p)),
               "gls_kernel_resources")
    return {"vgprs": v.value, "scratch_bytes": sc.value}


def gls_resident_capacity(n):
    return _lib.load().gnngls_gls_resident_capacity(int(n))


class gls_team_mode:
    """Experiment / test hook (gnngls_debug_set_gls_team): `with gls_team_mode(1): ...` runs gls_run with the team form of
    the perturbation phase wherever it exists, 0 never, -1 = the library's policy (B <= number of CUs).  Results are
    bit-identical either way."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        _lib.check(_lib.load().gnngls_debug_set_gls_team(self.mode), "debug_set_gls_team")
        return self

    def __exit__(self, *exc):
        _lib.check(_lib.load().gnngls_debug_set_gls_team(-1), "debug_set_gls_team")
        return False


class gls_prune_mode:
    """Experiment / test hook (gnngls_debug_set_gls_prune): `with gls_prune_mode(0): ...` makes the descent evaluate every
    move of its all-to-all scans instead of the pruned candidate sets (n >= 80).  Results are bit-identical either way."""

    def __init__(self, mode):
        self.mode = int(mode)

    def __enter__(self):
        _lib.check(_lib.load().gnngls_debug_set_gls_prune(self.mode), "debug_set_gls_prune")
        return self

    def __exit__(self, *exc):
        _lib.check(_lib.load().gnngls_debug_set_gls_prune(-1), "debug_set_gls_prune")
        return False


EXEC_RECORDS = 16        # GNNGLS_EXEC_RECORDS of include/gnngls_hip.h


class executed_evals:
    """Measurement hook (gnngls_profile_set_executed_evals): `with executed_evals(B_max) as x: gls_run(...)` -> x.counts
    [B_max] int64 holds, for the LAST gls_run launch inside the block, the delta evaluations the kernel actually executed
    per instance (<= GlsResult.evals, which counts what the reference evaluates; equal where no scan is pruned; -1 where
    the run pruned on a configuration without the counting instantiations -- see include/gnngls_hip.h).  A launch inside
    the block runs the COUNTING instantiation of the kernel (2-3 % slower): measure speed outside of it."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.buffer = torch.zeros((EXEC_RECORDS * self.capacity,), dtype=torch.int64, device=_dev())

    def record(self, B, k):
        """Record k of the LAST launch of B instances inside the block: 0 executed evaluations, 1 shader cycles of the
        workgroup, 2 shader cycles of its serial perturbation phase, 3 penalty steps of that phase, 4 100 MHz ticks, 5 .. 14 the
        cycle account of the descent (include/gnngls_hip.h, GNNGLS_EXEC_RECORDS)."""
        return self.buffer[k * B:(k + 1) * B]

    @property
    def counts(self):
        return self.buffer[:self.capacity]

    def __enter__(self):
        _lib.check(_lib.load().gnngls_profile_set_executed_evals(_lib.ptr(self.buffer)), "profile_set_executed_evals")
        return self

    def __exit__(self, *exc):
        _lib.check(_lib.load().gnngls_profile_set_executed_evals(None), "profile_set_executed_evals")
        return False


def gls_describe_config(n, B=0, penalty_bits=0, first_improvement=False):
    """-> dict(store, threads, lds_bytes, per_cu, team, waves_per_simd[, edge_form]): exactly what gnngls_gls_run launches for
    these arguments (host-side query, gnngls_gls_describe_run).  `edge_form` (the serial perturbation phase with the tour edges
    in registers) is reported under its own key by gls_describe_run()."""
    d = gls_describe_run(n, B, penalty_bits, first_improvement)
    d.pop("edge_form")
    return d


def gls_describe_run(n, B=0, penalty_bits=0, first_improvement=False):
    vals = [ctypes.c_int(0) for _ in range(7)]
    _lib.check(_lib.load().gnngls_gls_describe_run(int(n), int(B), int(penalty_bits), int(bool(first_improvement)),
                                                   *[ctypes.cast(ctypes.byref(v), ctypes.c_void_p) for v in vals]),
               "gls_describe_run")
    names = {0: "global", 116: "lds-tri-u16", 132: "lds-tri-i32", 200: "compact"}
    return {"store": names[vals[0].value], "threads": vals[1].value, "lds_bytes": vals[2].value, "per_cu": vals[3].value,
            "team": bool(vals[5].value), "waves_per_simd": vals[4].value, "edge_form": bool(vals[6].value)}
