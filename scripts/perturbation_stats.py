#!/usr/bin/env python
"""Which one-to-all scan of a penalty step accepts the step's first move (CPU oracle, diagnostics for the kernel design).
    python scripts/perturbation_stats.py n [instances] [outer_iters] [guide: weight|noise]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from gnngls_amd.synthetic import random_instances  # noqa: E402
from oracle import gls_oracle as go  # noqa: E402

n = int(sys.argv[1]); B = int(sys.argv[2]) if len(sys.argv) > 2 else 8; K = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
kind = sys.argv[4] if len(sys.argv) > 4 else "noise"
L = go.lib()
i64 = ctypes.c_int64
L.gls_oracle_perturbation_stats.argtypes = [ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.c_int]
L.gls_oracle_perturbation_stats(None, None, None, 1)
D, _ = random_instances(np.random.default_rng(0), B, n)
rng = np.random.default_rng(1)
for b in range(B):
    g = D[b]
    if kind == "noise":
        x = np.triu(np.maximum(rng.normal(0.05, 0.1, size=(n, n)).astype(np.float32).astype(np.float64), 0), 1)
        g = x + x.T
    init = go.nearest_neighbor(g)
    go.guided_local_search(D[b], g[None], init, go.tour_cost(init, D[b]), perturbation_moves=20, max_outer_iters=K, trace_cap=1, want_penalty=False)
h, m, st = (i64 * 5)(), (i64 * 4)(), i64()
L.gls_oracle_perturbation_stats(h, m, ctypes.byref(st), 0)
h, m = np.array(h[:], dtype=float), np.array(m[:], dtype=float)
print(f"n={n} guide={kind} steps={st.value} moves/step={m.sum() / st.value:.3f}")
print("first move of a step at scan [e0.2opt, e0.reloc, e1.2opt, e1.reloc, none]:", np.round(h / st.value, 3))
print("moves per step by scan:", np.round(m / st.value, 3))
