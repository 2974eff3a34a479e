#!/usr/bin/env python
"""Diagnostic (library built with GNNGLS_EXTRA_FLAGS=-DGLS_STAMPS): where the search kernel spends its cycles."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gnngls_amd import ops
from gnngls_amd.synthetic import random_instances
import os
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
D = torch.from_numpy(random_instances(np.random.default_rng(0), B, n)[0]).cuda()
init = ops.nearest_neighbor(D); cost = ops.tour_cost(init, D)
g = D[None].contiguous()
if len(sys.argv) > 3 and sys.argv[3] == "noise":
    x = np.maximum(np.random.default_rng(1).normal(0.05, 0.1, size=(B, n, n)).astype(np.float32).astype(np.float64), 0)
    x = np.triu(x, 1)
    g = torch.from_numpy((x + x.transpose(0, 2, 1))[None]).cuda().contiguous()
    init = ops.nearest_neighbor(g[0]); cost = ops.tour_cost(init, D)
if len(sys.argv) > 3 and sys.argv[3] == "model":          # the bench's guide: regret_pred of the synthetic (untrained) model
    from gnngls_amd import pipeline
    model = pipeline.synthetic_model(seed=1234)
    R = pipeline.predict_regret(model, D, pipeline.Scalers.fit_weights(D))
    g = R[None].contiguous()
    init = ops.nearest_neighbor(R); cost = ops.tour_cost(init, D)
# the stamp sink is a side buffer registered through the debug hook of the C ABI;
# trace_cap=0 = the throughput path (trace-free kernel instantiation, deferred tour_cost)
from gnngls_amd import _lib
stamps = torch.zeros((4 * B, 16), dtype=torch.int64, device="cuda")
_lib.check(_lib.load().gnngls_debug_set_stamp_buffer(_lib.ptr(stamps)))
tc = int(os.environ.get("TRACE_CAP", "0"))
_lib.check(_lib.load().gnngls_debug_set_gls_team(int(os.environ.get("TEAM", "-1"))))
_lib.check(_lib.load().gnngls_debug_set_gls_prune(int(os.environ.get("PRUNE", "-1"))))
print("config", ops.gls_describe_config(n, B, int(os.environ.get('BITS', '0'))))
r = ops.gls_run(D, g, init, cost, penalty_bits=int(os.environ.get('BITS', '0')), perturbation_moves=20, max_outer_iters=-1, time_limit_s=1.0,
                trace_cap=tc)
torch.cuda.synchronize()
per_wave = stamps[B:2 * B].double().mean(0).cpu().numpy()
scan_w = stamps[2 * B:3 * B].double().mean(0).cpu().numpy()
wait_w = stamps[3 * B:].double().mean(0).cpu().numpy()
stamps = stamps[:B]
raw = stamps.cpu().numpy()
st = stamps.double().mean(0).cpu().numpy()
assert st.sum() > 0, 'library was not built with GNNGLS_EXTRA_FLAGS=-DGLS_STAMPS'
names = ["utility argmax", "o2a scan (+pen, pos search)", "o2a reduce", "apply+reload", "phase tail", "descent (LS)", "steps"]
st[5] += st[8] + st[9] + st[10]          # the descent's sub-stamps restart the clock: slot 5 only holds the remainder
tot = st[:6].sum()
it = r.outer_iters.double().mean().item()
print(f"outer iters {it:.0f}, perturbation steps/iter {st[6] / it:.1f}")
for k in range(6):
    print(f"{names[k]:30s} {st[k] / tot * 100:5.1f}%   {st[k] / it:9.0f} cycles/outer-iter")
print(f"cycles per perturbation step (wave 0): {(st[0] + st[1] + st[2] + st[3] + per_wave[5] + per_wave[6]) / st[6]:.0f}")
if per_wave[5] > 0:
    print(f"edge form, per step: latch / loop top {per_wave[5] / st[6]:.0f}, utility divisions (+ wait for the last move's loads) {per_wave[6] / st[6]:.0f}, "
          f"arg-max reduction {st[0] / st[6]:.0f}, scans {st[1] / st[6]:.0f}, acceptance + reduction {st[2] / st[6]:.0f}, move {st[3] / st[6]:.0f}")
print(f"descent: scans/iter {st[11] / it:.2f}; per scan: scan {st[8] / st[11]:.0f}, arg-min+wait {st[9] / st[11]:.0f}, "
      f"apply+barrier {st[10] / st[11]:.0f} cycles (share of descent {100 * (st[8] + st[9] + st[10]) / max(st[5], 1):.0f}%)")
sc = max(st[11], 1)
print(f"descent per wave, cycles per scan: scan w0 {st[8] / sc:.0f}  w1 {st[13] / sc:.0f}  w2 {st[14] / sc:.0f}  w3 {st[15] / sc:.0f};  "
      f"arg-min+wait w0 {st[9] / sc:.0f}  w1 {st[12] / sc:.0f}")
simd = [((raw[:, 7] >> (8 * w)) & 0xff) - 1 for w in range(4)]
import collections
print("SIMD of waves 0..3 (count of instances):", collections.Counter(zip(*[x.tolist() for x in simd])).most_common(6))
print(f"moves/iter {r.trace_len.double().mean().item() / it:.1f}, evals/iter {r.evals.double().mean().item() / it:.0f}")
if per_wave[8] > 0:
    full = max(sc / 2 - per_wave[8], 1)
    print(f"quiet rows (wavefront 0): reduced relocate scans / iter {per_wave[8] / it:.2f}, cycles per refresh + reduced scan {per_wave[9] / per_wave[8]:.0f}, "
          f"flagged rows of the wavefront after a reduced scan {per_wave[11] / per_wave[8]:.1f}; full relocate scans / iter {full / it:.2f}, "
          f"cycles each {(per_wave[4] - per_wave[9]) / full:.0f}")
if n < 128 and per_wave[4] > 0:
    print(f"descent scans of wavefront 0: relocate {per_wave[4] / max(sc / 2, 1):.0f} cycles per scan, 2-opt {(st[8] - per_wave[4]) / max(sc / 2, 1):.0f}")
if int(os.environ.get("TEAM", "-1")) == 0 and per_wave[:4].sum() > 0:
    print(f"pruned scans (wavefront 0): overflow rows per scan 2-opt {per_wave[0] / max(sc / 2, 1):.2f}, relocate {per_wave[1] / max(sc / 2, 1):.2f}; "
          f"wave-passes per scan {per_wave[2] / max(sc / 2, 1):.2f} / {per_wave[3] / max(sc / 2, 1):.2f}")
elif per_wave.sum() > 0:
    print("team rounds: unit cycles per outer iteration, per wavefront:", " ".join(f"{v / it:.0f}" for v in per_wave))
print("descent, per wavefront, cycles per scan: scan      ", " ".join(f"{v / sc:.0f}" for v in scan_w))
print("descent, per wavefront, cycles per scan: arg-min+wait", " ".join(f"{v / sc:.0f}" for v in wait_w))
