"""PyTorch-ROCm custom operators of the hot path: `torch.ops.gnngls.*` (north_star).

Five operators are registered with `torch.library` over the C ABI of libgnngls_hip.so (include/gnngls_hip.h).  Each has
exactly ONE implementation, for the CUDA dispatch key (= HIP on ROCm): there is no CPU kernel behind any of them, so
calling one with CPU tensors fails in the dispatcher ("no CPU fallback" is structural, not a runtime check).  Shape
functions (fake/meta kernels) are registered so the ops can be traced and used under FakeTensorMode.  All ops enqueue on
the current HIP stream, take caller-owned contiguous tensors and retain nothing.

    regret_forward(feat[B,N] f32, packed_weights f32, n, heads=8, head_dim=16, hidden=512, layers) -> [B,N] f32
        EdgePropertyPredictionModel.forward (models.py:44-70); packed_weights = model.pack_weights(device)
    two_opt_delta_all(tour[B,n+1] i32, D[B,n,n] f64) -> [B,n+1,n+1] f64          operators.py:14-29
    relocate_delta_all(tour, D) -> [B,n+1,n+1] f64                                operators.py:83-103
    local_search(tour, cost[B] f64, D, first_improvement) -> (tour, cost, n_moves[B] i32)      algorithms.py:111-132
    gls_run(D, guides[G,B,n,n] f64, init_tour, init_cost, perturbation_moves, max_outer_iters, time_limit_s,
            first_improvement, trace_capacity) -> (best_tour, best_cost, outer_iters[B] i64, trace_cost[B,T] f64,
            trace_len[B] i32)                                                      algorithms.py:135-195
"""
import ctypes

import torch

from . import _lib, ops

_LIB = torch.library.Library("gnngls", "DEF")
_LIB.define("regret_forward(Tensor feat, Tensor packed_weights, int n, int heads, int head_dim, int hidden, int layers) -> Tensor")
_LIB.define("two_opt_delta_all(Tensor tour, Tensor D) -> Tensor")
_LIB.define("relocate_delta_all(Tensor tour, Tensor D) -> Tensor")
_LIB.define("local_search(Tensor tour, Tensor cost, Tensor D, bool first_improvement) -> (Tensor, Tensor, Tensor)")
_LIB.define("gls_run(Tensor D, Tensor guides, Tensor init_tour, Tensor init_cost, int perturbation_moves, "
            "int max_outer_iters, float time_limit_s, bool first_improvement, int trace_capacity) "
            "-> (Tensor, Tensor, Tensor, Tensor, Tensor)")

_workspaces = {}      # device index -> uint8 scratch tensor for the forward (grown on demand, reused across calls)


def _regret_forward(feat, packed_weights, n, heads, head_dim, hidden, layers):
    if (heads, head_dim, hidden) != (8, 16, 512):
        raise NotImplementedError("gnngls::regret_forward is specialised to the reference architecture "
                                  "(8 heads x 16, hidden 512: models.py:23,60)")
    L = _lib.load()
    N = n * (n - 1) // 2
    feat = feat.contiguous().float()
    in_dim = 1 if feat.dim() == 2 else feat.shape[-1]
    B = feat.numel() // (N * in_dim)
    expect = int(L.gnngls_model_packed_floats(in_dim, layers))
    if packed_weights.numel() != expect or packed_weights.dtype != torch.float32:
        raise ValueError(f"packed_weights must hold {expect} fp32 values for in_dim={in_dim}, layers={layers}")
    need = int(L.gnngls_regret_forward_workspace_bytes(B, n))
    ws_bytes = min(need, max(int(L.gnngls_regret_forward_workspace_bytes(1, n)), 48 << 30))
    key = feat.device.index
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < ws_bytes:
        _workspaces[key] = None
        ws = _workspaces[key] = torch.empty(ws_bytes, dtype=torch.uint8, device=feat.device)
    y = torch.empty((B, N), dtype=torch.float32, device=feat.device)
    _lib.check(L.gnngls_regret_forward(_lib.ptr(feat), _lib.ptr(packed_weights.contiguous()), B, n, in_dim, layers, _lib.ptr(y),
                                       _lib.ptr(ws), ctypes.c_int64(ws.numel()), _lib.current_stream()), "regret_forward")
    return y


def _local_search(tour, cost, D, first_improvement):
    r = ops.gls_run(D, None, tour, cost, first_improvement=first_improvement, max_outer_iters=0)
    return r.best_tour, r.best_cost, r.trace_len


def _gls_run(D, guides, init_tour, init_cost, perturbation_moves, max_outer_iters, time_limit_s, first_improvement,
             trace_capacity):
    r = ops.gls_run(D, guides, init_tour, init_cost, perturbation_moves=perturbation_moves,
                    first_improvement=first_improvement, max_outer_iters=max_outer_iters, time_limit_s=time_limit_s,
                    trace_cap=trace_capacity)
    trace = r.trace_cost if r.trace_cost is not None else torch.zeros((D.shape[0], 0), dtype=torch.float64, device=D.device)
    return r.best_tour, r.best_cost, r.outer_iters, trace, r.trace_len


_LIB.impl("regret_forward", _regret_forward, "CUDA")
_LIB.impl("two_opt_delta_all", ops.two_opt_delta_all, "CUDA")
_LIB.impl("relocate_delta_all", ops.relocate_delta_all, "CUDA")
_LIB.impl("local_search", _local_search, "CUDA")
_LIB.impl("gls_run", _gls_run, "CUDA")


# ---- shape functions (fake tensors / tracing); no arithmetic -------------------------------------------------------
@torch.library.register_fake("gnngls::regret_forward")
def _(feat, packed_weights, n, heads, head_dim, hidden, layers):
    N = n * (n - 1) // 2
    in_dim = 1 if feat.dim() == 2 else feat.shape[-1]
    return feat.new_empty((feat.numel() // (N * in_dim), N), dtype=torch.float32)


def _table_shape(tour, D):
    B, n1 = tour.shape
    return D.new_empty((B, n1, n1), dtype=torch.float64)


torch.library.register_fake("gnngls::two_opt_delta_all")(_table_shape)
torch.library.register_fake("gnngls::relocate_delta_all")(_table_shape)


@torch.library.register_fake("gnngls::local_search")
def _(tour, cost, D, first_improvement):
    return torch.empty_like(tour), torch.empty_like(cost), tour.new_empty((tour.shape[0],), dtype=torch.int32)


@torch.library.register_fake("gnngls::gls_run")
def _(D, guides, init_tour, init_cost, perturbation_moves, max_outer_iters, time_limit_s, first_improvement, trace_capacity):
    B = D.shape[0]
    return (torch.empty_like(init_tour), torch.empty_like(init_cost), D.new_empty((B,), dtype=torch.int64),
            D.new_empty((B, trace_capacity), dtype=torch.float64), D.new_empty((B,), dtype=torch.int32))
