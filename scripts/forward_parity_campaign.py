#!/usr/bin/env python
"""Randomised parity campaign of the inference forward: random instance sizes, batch sizes and synthetic checkpoints
through the same check as tests/test_model_gpu.py::test_forward_batch_vs_oracle (1e-5 relative on the regret predictions
against the fp64 oracle, with the fp32-reference clause for ill-conditioned tiny graphs).

    python scripts/forward_parity_campaign.py [--cases 40] [--seed 1] [--max_n 60]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_model_gpu as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max_n", type=int, default=60)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0, failed = time.time(), []
    for _ in range(args.cases):
        n, B = int(rng.integers(3, args.max_n + 1)), int(rng.integers(1, 5))
        try:
            T.test_forward_batch_vs_oracle(n, B)
        except AssertionError as e:
            failed.append((n, B, str(e)[:160]))
    print(f"{args.cases} random (n, batch) forwards checked at 1e-5 against the fp64 oracle, {len(failed)} failures, "
          f"{time.time() - t0:.0f} s")
    print(f"cases that needed the small-graph clause (n < 20, 3x the fp32 reference's own error): "
          f"{sorted(set((n, i) for n, i, _, _ in T.ESCAPES))}")
    for f in failed:
        print("FAIL", f)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
