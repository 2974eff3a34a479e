#!/usr/bin/env python
"""Golden outputs of the CPU model oracle (oracle/model_oracle.py evaluated in fp64) at the sizes the bench runs -- n = 100 and
n = 200, 16 instances each over four seeded checkpoints (two at initialisation scale, two with trained-like weight scales:
BatchNorm gamma up to 4, calibrated running statistics, GATConv fc at 3 x and 1 x its initial gain) --
for tests/test_model_gpu.py::test_forward_error_fixtures and scripts/forward_error_campaign.py.  An fp64 oracle forward takes
11 s at n = 100 and ~2 min at n = 200 on the build container's CPUs, which is why these are fixtures (data: seeds, expected
outputs, and the error of a plain fp32 evaluation of the same graph for comparison) and not computed on the GPU box.

    python tests/golden/make_forward_error_fixtures.py 100 200
"""
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model_oracle as mo  # noqa: E402

CHECKPOINTS = [(1234, 99, 0), (4321, 7, 0), (1111, 21, 3), (2222, 22, 1)]      # (model seed, state-dict seed, 0 = initialisation scale, else trained-like with this fc gain)
PER_CKPT = 4


def checkpoint(model_seed, sd_seed, kind, running_stats=None):
    """-> (oracle in eval mode, state dict, running statistics of a trained-like checkpoint or None)"""
    torch.manual_seed(model_seed)
    oracle = mo.EdgeRegretModelOracle(1, 128, 1, 3, n_heads=8)
    stats = None
    if kind:
        sd, stats = mo.trained_like_state_dict(oracle, sd_seed, fc_gain=float(kind), running_stats=running_stats)
    else:
        sd = mo.synthetic_state_dict(oracle, sd_seed)
    oracle.load_state_dict(sd)
    return oracle.eval(), sd, stats


def features(n, ckpt, k):
    N = n * (n - 1) // 2
    return np.random.default_rng([n, ckpt, k]).random((N, 1)).astype(np.float32)


def main():
    for n in [int(a) for a in sys.argv[1:]]:
        N = n * (n - 1) // 2
        G = mo.line_graph_arcs_closed_form(n)
        ref64 = np.zeros((len(CHECKPOINTS), PER_CKPT, N))
        err32 = np.zeros((len(CHECKPOINTS), PER_CKPT))
        t0 = time.time()
        stats_out = {}
        for c, (ms, ss, kind) in enumerate(CHECKPOINTS):
            oracle, _, stats = checkpoint(ms, ss, kind)
            if stats is not None:                                # calibrated BatchNorm statistics: stored, so that every machine rebuilds the same checkpoint
                for key, v in stats.items():
                    stats_out[f"stats{c}:{key}"] = v.numpy()
            o64 = copy.deepcopy(oracle).double()
            for k in range(PER_CKPT):
                x = torch.from_numpy(features(n, c, k))
                with torch.no_grad():
                    r64 = o64(G, x.double()).numpy().reshape(-1)
                    r32 = oracle(G, x).numpy().reshape(-1).astype(np.float64)
                ref64[c, k] = r64
                err32[c, k] = np.abs(r32 - r64).max()
                print(f"n={n} checkpoint {c} instance {k}: max|y| {np.abs(r64).max():.4g}, fp32 evaluation max err {err32[c, k]:.3e} "
                      f"= {err32[c, k] / (1e-5 * np.abs(r64).max()):.2f} of 1e-5 max|y|  ({time.time() - t0:.0f} s)", flush=True)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"forward_error_n{n}.npz"), ref64=ref64, fp32_eval_max_err=err32,
                            checkpoints=np.array(CHECKPOINTS), per_checkpoint=PER_CKPT, **stats_out)


if __name__ == "__main__":
    main()
