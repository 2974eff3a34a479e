"""GPU smoke tests of bench.py itself: the single-rank line and the multi-rank path (2 ranks sharing one GPU over
gloo -- the only difference to the driver's torchrun launch is the backend of the final gather)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json_line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert lines, out.decode()[-2000:]
    return json.loads(lines[-1])


def test_bench_single_rank_small():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                                   "--batch", "64", "--time_limit", "0.5"], cwd=ROOT, stderr=subprocess.STDOUT, timeout=600)
    j = last_json_line(out)
    assert j["n_gpus"] == 1 and j["steps"] == 1 and j["unit"] == "instances/s"
    assert 64 / 1.5 < j["value"] < 64 / 0.45
    assert j["roofline"]["bound"] in ("hbm", "mfma") and j["roofline"]["traffic"] is not None
    assert j["cpu_baseline"]["cores"] >= 1 and j["cpu_baseline"]["kind"] == "port"
    assert j["watchdog_aborts"] == 0 and j["roofline_gls"]["launches"] == 1


def test_bench_two_ranks_gloo():
    env = dict(os.environ, GNNGLS_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
           "--batch", "64", "--time_limit", "0.5"]
    out = subprocess.check_output(cmd, cwd=ROOT, env=env, stderr=subprocess.STDOUT, timeout=600)
    j = last_json_line(out)
    assert j["n_gpus"] == 2 and j["scaling"] == "weak"
    assert 128 / 2.5 < j["value"] < 128 / 0.45          # whole-job aggregate over both ranks
    assert "cpu_baseline" not in j                      # rank 0, N=1 only
