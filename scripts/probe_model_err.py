#!/usr/bin/env python
"""Diagnostic: error of the HIP fp32 forward and of the fp32 CPU oracle against the fp64 oracle."""
import copy, sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import model_oracle as mo
from gnngls_amd.models import EdgePropertyPredictionModel, LineGraph
torch.manual_seed(1234)
oracle = mo.EdgeRegretModelOracle(1, 128, 1, 3, n_heads=8)
sd = mo.synthetic_state_dict(oracle, seed=99); oracle.load_state_dict(sd); oracle.eval()
o64 = copy.deepcopy(oracle).double()
model = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8); model.load_state_dict(sd); model.eval().to("cuda")
for n, B in [(3, 4), (4, 3), (5, 2), (10, 2), (20, 2), (33, 1), (50, 1), (100, 1)]:
    N = n * (n - 1) // 2
    rng = np.random.default_rng(n)
    x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
    G = mo.line_graph_networkx(n)
    with torch.no_grad():
        yh = model(LineGraph(n, batch=B).to("cuda"), x.cuda()).cpu().double().reshape(B, N)
        for b in range(B):
            y32 = oracle(G, x[b * N:(b + 1) * N]).double().reshape(-1)
            y64 = o64(G, x[b * N:(b + 1) * N].double()).reshape(-1)
            s = y64.abs().max().item()
            print(f"n={n} b={b} max|y|={s:.3f}  ref32-vs-64 {(y32 - y64).abs().max().item() / s:.2e}  "
                  f"hip-vs-64 {(yh[b] - y64).abs().max().item() / s:.2e}  hip-vs-ref32 {(yh[b] - y32).abs().max().item() / s:.2e}")
