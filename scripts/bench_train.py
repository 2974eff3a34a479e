#!/usr/bin/env python
"""Training-step throughput of the HIP path (the step of scripts/train.py:20-32: forward in training
mode, MSELoss, backward, Adam) on a synthetic dgl.batch-shaped batch.  Prints one JSON line.

    python scripts/bench_train.py [--n 100] [--batch 32] [--steps 20] [--warmup 3]

FLOP model per instance and layer (N = n(n-1)/2 line-graph nodes, E = N * 2(n-2) arcs):
    forward  : fc 2*N*128^2 + attention E*304 + MLP 4*N*128*512
    backward : data gradients (fc 2*N*128^2, MLP 4*N*128*512 + the recomputed first MLP GEMM 2*N*128*512),
               weight gradients (fc 2*N*128^2, MLP 4*N*128*512), attention backward E*(2*16 + 2*16 + 12)*8/8 ~ E*608
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnngls_amd import _lib, models  # noqa: E402


def flops_per_instance(n, layers=8):
    N = n * (n - 1) // 2
    E = N * 2 * (n - 2)
    fc, mlp = 2 * N * 128 * 128, 4 * N * 128 * 512
    fwd = fc + E * 304 + mlp
    bwd = (fc + mlp + mlp // 2) + (fc + mlp) + E * 608
    return layers * fwd, layers * bwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-optimizer", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step (forward, loss, backward, Adam) in one HIP graph and replay it")
    args = ap.parse_args()
    n, B = args.n, args.batch
    N = n * (n - 1) // 2
    torch.manual_seed(0)
    model = models.EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8).cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, capturable=args.graph)   # train.py:104
    crit = torch.nn.MSELoss()                                                   # train.py:108
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32)).cuda()
    G = models.LineGraph(n, batch=B).to("cuda")

    def step():
        opt.zero_grad(set_to_none=not args.graph)
        loss = crit(model(G, x), y)
        loss.backward()
        if not args.no_optimizer:
            opt.step()
        return loss

    if args.graph:
        # torch.cuda.graphs "whole network capture": every launch the C ABI enqueues on torch's current stream is recorded
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(args.warmup, 3)):
                step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=False)
        with torch.cuda.graph(graph):
            static_loss = step()
        run = lambda: (graph.replay(), static_loss)[1]  # noqa: E731
    else:
        run = step
        for _ in range(args.warmup):
            step()
    torch.cuda.synchronize()
    _lib.profile_enable(not args.graph)
    t0 = time.time()
    for _ in range(args.steps):
        loss = run()
    torch.cuda.synchronize()
    dt = time.time() - t0
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    fwd, bwd = flops_per_instance(n)
    ms = dt / args.steps * 1e3
    kern = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps} for k, v in prof.items() if v[1]}
    # algorithmic FLOP per step of the MFMA kernel classes (8 layers, M = B*N rows) -> achieved TFLOP/s vs the 157.3 dense
    # fp32 MFMA peak of MI355X
    M, E = B * N, B * N * 2 * (n - 2)
    algo = {"ffn_fused": 8 * 4 * M * 128 * 512, "gemm_fc": 8 * 2 * M * 128 * 128,
            "train_gemm_bwd": 8 * (4 * M * 128 * 512 + 2 * M * 128 * 128),
            "train_gemm_tn": 8 * (4 * M * 128 * 512 + 2 * M * 128 * 128),
            "gat_rows": 8 * E * 304, "train_gat_bwd": 8 * E * 608}
    for k, f in algo.items():
        if k in kern:
            kern[k]["tflops"] = f / (kern[k]["ms_per_step"] * 1e-3) / 1e12
            kern[k]["frac_of_mfma_peak"] = kern[k]["tflops"] / 157.3
    roofline = None
    if kern:
        top = max((k for k in algo if k in kern), key=lambda k: kern[k]["ms_per_step"])
        roofline = {"bound": "mfma", "kernel": top, "achieved": kern[top]["tflops"], "peak": 157.3, "unit": "TFLOP/s",
                    "frac": kern[top]["tflops"] / 157.3, "traffic": None}
    out = {"metric": "training steps/sec (forward+backward+Adam)", "value": args.steps / dt, "unit": "steps/s",
           "instances_per_s": B * args.steps / dt, "ms_per_step": ms, "n": n, "batch": B, "steps": args.steps,
           "warmup": args.warmup, "hip_graph": bool(args.graph), "dtype": "f32", "data": "synthetic", "loss": float(loss.item()),
           "model_tflops": (fwd + bwd) * B / (ms * 1e-3) / 1e12,
           "kernel_ms_per_step": sum(v["ms_per_step"] for v in kern.values() if "ms_per_step" in v), "roofline": roofline,
           "kernels": kern,
           "workspace_gib": _lib.load().gnngls_regret_train_workspace_bytes(B, n, 8) / 2 ** 30}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n)
    print(json.dumps(out))


def cpu_baseline(n):
    """The CPU oracle (plain PyTorch fp32 autograd of the reference's graph, oracle/model_oracle.py) on a bounded sample:
    one forward+backward+Adam step on a batch of ONE instance of the same size, all host cores."""
    from oracle import model_oracle as mo
    torch.manual_seed(0)
    model = mo.EdgeRegretModelOracle(1, 128, 1, 3, n_heads=8).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    N = n * (n - 1) // 2
    G = mo.batch_line_graphs(n, 1)
    x, y = torch.rand(N, 1), torch.rand(N, 1)
    t0 = time.time()
    opt.zero_grad()
    loss = torch.nn.functional.mse_loss(model(G, x), y)
    loss.backward()
    opt.step()
    dt = time.time() - t0
    return {"value": 1.0 / dt, "unit": "instances/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"one training step on a batch of 1 TSP{n} instance (fp32 torch autograd of the oracle graph)",
            "wall_s": dt}


if __name__ == "__main__":
    main()
