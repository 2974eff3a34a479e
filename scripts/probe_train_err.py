"""Diagnostic: per-tensor gradient error of the HIP training step and of the fp32 CPU oracle, both against the fp64 oracle.
usage: python scripts/probe_train_err.py n B"""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import model_oracle as mo  # noqa: E402
import test_train_gpu as T  # noqa: E402

n, B = int(sys.argv[1]), int(sys.argv[2])
model, oracle = T.make_models(4321, 77)
oracle64 = copy.deepcopy(oracle).double()
N = n * (n - 1) // 2
rng = np.random.default_rng(100 * n + B)
x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
t = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
G = mo.batch_line_graphs(n, B)
y32, l32, g32, _ = mo.train_step_reference(oracle, G, x, t)
y64, l64, g64, _ = mo.train_step_reference(oracle64, G, x.double(), t.double())
y, loss, grads, _ = T.hip_step(model, n, B, x, t)
print("pred err hip %.3e fp32 %.3e" % ((y.double() - y64).abs().max().item(), (y32.double() - y64).abs().max().item()))
for k, r in g64.items():
    m = r.abs().max().item()
    eh = (grads[k].double() - r).abs().max().item()
    e3 = (g32[k].double() - r).abs().max().item()
    print(f"{k:62s} max {m:.2e} hip {eh:.2e} ({eh / max(m, 1e-30):.1e}) fp32 {e3:.2e} ({e3 / max(m, 1e-30):.1e}) ratio {eh / max(e3, 1e-30):.1f}")
rh = [(grads[k].double() - r).abs().max().item() / r.abs().max().item() for k, r in g64.items() if r.abs().max().item() > 1e-9]
r3 = [(g32[k].double() - r).abs().max().item() / r.abs().max().item() for k, r in g64.items() if r.abs().max().item() > 1e-9]
print("SUMMARY n=%d B=%d  hip: median rel %.2e worst %.2e   fp32 oracle: median rel %.2e worst %.2e" %
      (n, B, np.median(rh), max(rh), np.median(r3), max(r3)))
