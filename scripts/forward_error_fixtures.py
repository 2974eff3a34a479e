#!/usr/bin/env python
"""The inference forward against the fp64 oracle at the sizes the bench runs (n = 100 and n = 200, 16 instances each, four seeded
checkpoints: two at initialisation scale, two with trained-like weight scales), from the committed fixtures
(tests/golden/forward_error_n{100,200}.npz, made by tests/golden/make_forward_error_fixtures.py on the build container's CPUs).

    python scripts/forward_error_fixtures.py [100 200]          (GNNGLS_FFN_FP32=1: the feed-forward block on the fp32 pipe)

Per instance: the worst |y - ref64| as a multiple of the bar in the form SURVEY 7.4-6 writes it, 1e-5 max(|ref|, max|ref|) = 1e-5
max|ref|, and in the looser sum form earlier rounds asserted (1e-5 |ref| + 1e-5 max|ref|); next to it the same figure for a plain
fp32 evaluation of the reference's own graph (oracle/model_oracle.py in fp32 on the CPU), which the fixtures recorded.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_forward_error_fixtures as F  # noqa: E402


def run(n):
    from gnngls_amd import models as M
    fx = np.load(os.path.join(ROOT, "tests", "golden", f"forward_error_n{n}.npz"))
    N = n * (n - 1) // 2
    rows = []
    for c, (ms, ss, kind) in enumerate(fx["checkpoints"].tolist()):
        stats = {k.split(":", 1)[1]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith(f"stats{c}:")} or None
        _, sd, _ = F.checkpoint(ms, ss, kind, stats)
        model = M.EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8)
        model.load_state_dict(sd)
        model.eval().to("cuda")
        per = int(fx["per_checkpoint"])
        x = torch.from_numpy(np.concatenate([F.features(n, c, k) for k in range(per)])).cuda()
        with torch.no_grad():
            y = M.regret_forward(model, x, per, n).cpu().numpy().astype(np.float64)
        for k in range(per):
            ref = fx["ref64"][c, k]
            err = np.abs(y[k] - ref)
            scale = np.abs(ref).max()
            rows.append({"n": n, "checkpoint": c, "kind": kind, "instance": k, "max_abs_ref": float(scale),
                         "hip_over_strict_bar": float(err.max() / (1e-5 * scale)),
                         "hip_over_sum_bar": float((err / (1e-5 * np.abs(ref) + 1e-5 * scale)).max()),
                         "fp32_eval_over_strict_bar": float(fx["fp32_eval_max_err"][c, k] / (1e-5 * scale))})
    return rows


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [100, 200]
    path = "fp32 MFMA (GNNGLS_FFN_FP32=1)" if os.environ.get("GNNGLS_FFN_FP32", "0") not in ("", "0") else "bf16 MFMA, three pieces per operand"
    print(f"feed-forward block on: {path}")
    print("  n ckpt kind inst   max|ref|   HIP / strict bar   HIP / sum-form bar   fp32 evaluation of the reference graph / strict bar")
    allrows = []
    for n in sizes:
        allrows += run(n)
    for r in allrows:
        print(f"{r['n']:4d} {r['checkpoint']:4d} {r['kind']:4d} {r['instance']:4d} {r['max_abs_ref']:10.4g} {r['hip_over_strict_bar']:18.3f} "
              f"{r['hip_over_sum_bar']:20.3f} {r['fp32_eval_over_strict_bar']:20.3f}")
    h = np.array([r["hip_over_strict_bar"] for r in allrows]); f = np.array([r["fp32_eval_over_strict_bar"] for r in allrows])
    hs = np.array([r["hip_over_sum_bar"] for r in allrows])
    print(f"instances: {len(allrows)}; above the strict bar: HIP {int((h > 1).sum())}, fp32 evaluation {int((f > 1).sum())}; "
          f"pass the sum form only (strict > 1 >= sum): {int(((h > 1) & (hs <= 1)).sum())}; "
          f"HIP above the strict bar where the fp32 evaluation is within it: {int(((h > 1) & (f <= 1)).sum())}; "
          f"median HIP / fp32-evaluation error ratio {np.median(h / f):.2f}, max {np.max(h / f):.2f}")


if __name__ == "__main__":
    main()
