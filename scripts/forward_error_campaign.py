#!/usr/bin/env python
"""How far the inference forward is from the fp64 oracle, as a fraction of the parity bar (1e-5 |ref| + 1e-5 max|ref|), over random
(n, batch) cases and synthetic checkpoints -- next to the same figure for a plain fp32 evaluation of the reference's own graph
(oracle/model_oracle.py in fp32).  Run once as it is (feed-forward block on the bf16 pipe in three pieces per operand) and once with
GNNGLS_FFN_FP32=1 (fp32 pipe): the two columns must look alike.

    python scripts/forward_error_campaign.py [--cases 30] [--seed 5] [--min_n 20] [--max_n 70]
"""
import argparse
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--min_n", type=int, default=20)
    ap.add_argument("--max_n", type=int, default=70)
    args = ap.parse_args()
    from gnngls_amd.models import LineGraph
    from oracle import model_oracle as mo
    rng = np.random.default_rng(args.seed)
    hip, ref = [], []
    t0 = time.time()
    for case in range(args.cases):
        n, B = int(rng.integers(args.min_n, args.max_n + 1)), int(rng.integers(1, 4))
        model, oracle, _ = T.make_models(seed=1000 + case, sd_seed=2000 + case)
        oracle64 = copy.deepcopy(oracle).double()
        N = n * (n - 1) // 2
        x = torch.from_numpy(rng.random((B * N, 1)).astype(np.float32))
        with torch.no_grad():
            y = model(LineGraph(n, batch=B).to("cuda"), x.cuda()).cpu().numpy().reshape(B, N).astype(np.float64)
            G1 = mo.line_graph_networkx(n)
            for b in range(B):
                xb = x[b * N:(b + 1) * N]
                r64 = oracle64(G1, xb.double()).numpy().reshape(-1)
                r32 = oracle(G1, xb).numpy().reshape(-1).astype(np.float64)
                bound = T.RTOL * np.abs(r64) + T.RTOL * np.abs(r64).max()
                hip.append(float((np.abs(y[b] - r64) / bound).max()))
                ref.append(float((np.abs(r32 - r64) / bound).max()))
    hip, ref = np.array(hip), np.array(ref)
    path = "fp32 MFMA (GNNGLS_FFN_FP32=1)" if os.environ.get("GNNGLS_FFN_FP32", "0") not in ("", "0") else "bf16 MFMA, three pieces per operand"
    print(f"feed-forward block on: {path}; {args.cases} cases, {len(hip)} instances, n in [{args.min_n}, {args.max_n}], {time.time() - t0:.0f} s")
    print(f"  HIP forward, worst error / bar per instance:      median {np.median(hip):.3f}  p90 {np.quantile(hip, 0.9):.3f}  max {hip.max():.3f}")
    print(f"  fp32 reference evaluation, the same figure:        median {np.median(ref):.3f}  p90 {np.quantile(ref, 0.9):.3f}  max {ref.max():.3f}")
    print(f"  instances above the bar: HIP {int((hip > 1.0).sum())}, fp32 reference evaluation {int((ref > 1.0).sum())}; HIP above the bar where the fp32 reference evaluation is within half of it: "
          f"{int(((hip > 1.0) & (ref <= 0.5)).sum())}")


if __name__ == "__main__":
    main()
