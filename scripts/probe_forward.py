#!/usr/bin/env python
"""Forward-only timing on the GPU box: per-kernel-class device time (HIP events inside the C ABI)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from gnngls_amd import _lib, pipeline  # noqa: E402
from gnngls_amd.synthetic import random_instances  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
D = torch.from_numpy(random_instances(np.random.default_rng(0), B, n)[0]).cuda()
model = pipeline.synthetic_model()
sc = pipeline.Scalers.fit_weights(D)
pipeline.predict_regret(model, D, sc)
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(reps):
    pipeline.predict_regret(model, D, sc)
torch.cuda.synchronize()
prof = _lib.profile_collect()
_lib.profile_enable(False)
N = n * (n - 1) // 2
M = B * N
tot = 0.0
for k, (ms, cnt) in prof.items():
    if cnt == 0:
        continue
    avg = ms / cnt
    tot += ms / reps
    extra = ""
    if k == "gemm_ffn1" or k == "gemm_ffn2":
        extra = f"  {2.0 * M * 128 * 512 / (avg * 1e-3) / 1e12:.1f} TFLOP/s"
    if k == "gemm_fc":
        extra = f"  {2.0 * M * 128 * 128 / (avg * 1e-3) / 1e12:.1f} TFLOP/s"
    if k == "ffn_fused":
        extra = f"  {4.0 * M * 128 * 512 / (avg * 1e-3) / 1e12:.1f} TFLOP/s"
    if k == "gat_rows":
        extra = f"  {1600.0 * M / (avg * 1e-3) / 1e9:.0f} GB/s (K1 algorithmic bytes over this kernel alone)"
    print(f"{k:18s} launches/fwd={cnt // reps:3d} avg={avg:8.3f} ms{extra}")
print(f"forward total {tot:.1f} ms for {B} TSP{n} instances = {tot / B:.3f} ms/instance")
