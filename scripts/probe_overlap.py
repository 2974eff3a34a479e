#!/usr/bin/env python
"""Can the GNN forward of the NEXT device load run under the search of the current one?  (round-5 review, item 5b)

The search kernel's workgroups hold 40 KB of LDS and 4 wavefronts x 128 VGPRs each for the whole search (compact store, forced
here for every R); `gat_rows` needs 70 KB per workgroup, the feed-forward block 142 KB (GNNGLS_FFN_FP32=1: the fp32 form, 74 KB, the
only one that fits beside two search workgroups per CU).  This probe launches a 2 s search of R resident instances on one stream
and, on a second stream, forward passes of 1,024 instances, and reports what each side got done alone and together.

    GNNGLS_FFN_FP32=1 python scripts/probe_overlap.py            (on an MI355X)
"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from gnngls_amd import ops, pipeline  # noqa: E402
from gnngls_amd.synthetic import random_instances  # noqa: E402

n, T = 100, 2.0
Dall = torch.from_numpy(random_instances(np.random.default_rng(0), 1024, n)[0]).cuda()
model = pipeline.synthetic_model(seed=1234)
sc = pipeline.Scalers.fit_weights(Dall)
R = pipeline.predict_regret(model, Dall, sc)
torch.cuda.synchronize()


def search(B, stream):
    with torch.cuda.stream(stream):
        D, g = Dall[:B].contiguous(), R[:B][None].contiguous()
        init = ops.nearest_neighbor(R[:B].contiguous()); cost = ops.tour_cost(init, D)
        return ops.gls_run(D, g, init, cost, perturbation_moves=20, max_outer_iters=-1, time_limit_s=T,
                           penalty_bits=-2)      # the compact store at every R: 40 KB of LDS, 4 wavefronts at 128 VGPRs per instance


def forwards(stream, seconds):
    done, t0 = 0, time.time()
    with torch.cuda.stream(stream):
        while time.time() - t0 < seconds:
            pipeline.predict_regret(model, Dall, sc)
            stream.synchronize()
            done += 1
    return done, time.time() - t0


# --priority: the forward stream at high priority (a hardware queue of its own)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=-1 if "--priority" in sys.argv else 0)
k, dt = forwards(s2, 1.0)
print(f"forward alone: {dt / k * 1e3:.1f} ms per 1,024 instances")
for B in (1024, 768, 512, 256, 128):    # search workgroups per CU: 4, 3, 2, 1, one on every other CU
    r = search(B, s1); s1.synchronize()
    alone = float(r.outer_iters.double().mean())
    torch.cuda.synchronize()
    t0 = time.time()
    # ops.gls_run looks at the status words when the kernel is done (a host synchronisation): the search goes on a thread of its own
    box = {}
    th = threading.Thread(target=lambda: box.update(r=search(B, s1)))
    th.start()
    time.sleep(0.05)                       # let the search kernel start first
    k, dt = forwards(s2, T)
    th.join()
    s1.synchronize()
    r = box["r"]
    wall = time.time() - t0
    both = float(r.outer_iters.double().mean())
    print(f"search of {B} resident instances, {T:g} s: {alone:.0f} outer iterations alone, {both:.0f} with forward passes pushed on a second stream; "
          f"{k} forward passes completed in {dt:.2f} s ({dt / max(k, 1) * 1e3:.0f} ms each); search + forwards together took {wall:.2f} s of wall "
          f"({T:g} s = fully overlapped, {T + dt:.1f} s = one after the other)")
