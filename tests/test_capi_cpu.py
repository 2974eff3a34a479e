"""CPU checks of the boundary: the C-ABI library loads and exports every symbol include/gnngls_hip.h
declares, host-side queries work without a GPU, and compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gnngls_amd import _lib, build
    build.build()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gnngls_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnngls_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from gnngls_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 15
    raw = ctypes.CDLL(_lib.SO)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/gnngls_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in gnngls_amd/_lib.py"


def test_host_queries(lib):
    assert lib.gnngls_abi_version() == 4
    assert lib.gnngls_gls_resident_capacity(2) == 0
    caps = [lib.gnngls_gls_resident_capacity(n) for n in (20, 50, 100, 150)]
    assert all(c > 0 for c in caps) and caps == sorted(caps, reverse=True)
    assert lib.gnngls_gls_resident_capacity(400) == 0            # triangles exceed 160 KiB of LDS
    assert lib.gnngls_gls_resident_capacity(100) == 4 * 256       # compact store: 4 workgroups per CU
    n_layers = 8
    per_layer = 128 * 128 + 4 * 128 + 512 * 128 + 512 + 128 * 512 + 3 * 128
    assert lib.gnngls_model_packed_floats(1, n_layers) == 128 + 128 + n_layers * per_layer + 128 + 4
    assert lib.gnngls_regret_forward_workspace_bytes(2, 100) > 2 * 4950 * 128 * 4 * 5


def test_training_entry_points_host_side(lib):
    """Workspace query grows with batch, size and depth; argument checks fire before any HIP call (no GPU here)."""
    w = lib.gnngls_regret_train_workspace_bytes
    assert w(0, 100, 8) == 0 and w(1, 100, 8) > 0
    assert w(2, 100, 8) > w(1, 100, 8) and w(1, 100, 8) > w(1, 50, 8) and w(1, 100, 8) > w(1, 100, 4)
    rows = 4950
    assert w(1, 100, 8) >= rows * 4 * (9 * 128 + 8 * (4 * 128 + 512 + 16))        # saved activations alone
    assert lib.gnngls_regret_train_forward(None, None, 1, 100, 1, 8, 1e-5, None, None, None, 0, None) == -1
    assert b"regret_train_forward" in lib.gnngls_last_error()
    assert lib.gnngls_regret_train_backward(None, None, None, 1, 100, 1, 8, None, None, 0, None) == -1
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.gnngls_regret_train_forward(p, p, 1, 258, 1, 8, 1e-5, p, p, p, 1 << 40, None) == -3      # n > 257
    assert b"tile limit" in lib.gnngls_last_error()
    assert lib.gnngls_regret_train_forward(p, p, 1, 100, 1, 8, 1e-5, p, p, p, 1024, None) == -1         # workspace too small
    assert b"workspace too small" in lib.gnngls_last_error()


def test_bad_arguments_are_rejected(lib):
    from gnngls_amd import _lib
    assert lib.gnngls_tour_cost(None, None, 1, 5, None, None) == -1
    assert b"tour_cost" in lib.gnngls_last_error()
    with pytest.raises(_lib.GnnglsHipError):
        _lib.check(lib.gnngls_gls_run(None, None, 0, 1, 5, None, None, 20, 0, 0, 0, 0.0, 1.0, None, None, None, None, None,
                                      0, None, None, None, None, None, None, None, 0, None, None), "gls_run")


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gnngls_amd import _lib, ops
    with pytest.raises(_lib.GnnglsHipError):
        ops.as_dev([[0, 1, 0]], torch.int32)
    from gnngls_amd.models import EdgePropertyPredictionModel, LineGraph
    m = EdgePropertyPredictionModel(1, 128, 1, 3, n_heads=8).eval()
    with pytest.raises(_lib.GnnglsHipError):
        m(LineGraph(4), torch.zeros(6, 1))


def test_gls_store_selection(lib):
    """Fastest store that keeps the batch resident; the compact store is exactly 40 KiB at n=100 (4 per CU, four
    4-wave workgroups on the 128-VGPR build)."""
    from gnngls_amd import ops
    c = ops.gls_describe_config(100, 1024)
    assert c == {"store": "compact", "threads": 256, "lds_bytes": 40960, "per_cu": 4, "team": False, "waves_per_simd": 4}
    # perturbation phase on all wavefronts only where every workgroup of the batch owns a CU (B <= 256 CUs) and is a
    # 16-wave workgroup (one per CU by its LDS footprint: n >= 144)
    # round 5: ... and only for first-improvement runs (best improvement: the edge form of the serial phase is faster there too)
    t200 = ops.gls_describe_config(200, 256, first_improvement=True)
    assert t200["store"] == "compact" and t200["team"] and t200["threads"] == 1024 and t200["lds_bytes"] <= 160 * 1024
    b200 = ops.gls_describe_run(200, 256)
    assert b200["store"] == "compact" and not b200["team"] and b200["edge_form"] and b200["threads"] == 1024 and b200["waves_per_simd"] == 4
    assert not ops.gls_describe_run(200, 256, first_improvement=True)["edge_form"]
    assert ops.gls_describe_run(100, 1024)["edge_form"] and ops.gls_describe_run(20, 1000)["edge_form"] and ops.gls_describe_run(50, 128)["edge_form"]
    assert not ops.gls_describe_run(100, 700, penalty_bits=16)["edge_form"] and not ops.gls_describe_run(300, 8)["edge_form"]
    fi = dict(first_improvement=True)
    assert not ops.gls_describe_config(200, 257, **fi)["team"] and not ops.gls_describe_config(50, 128, **fi)["team"]
    assert not ops.gls_describe_config(100, 256, **fi)["team"] and not ops.gls_describe_config(130, 8, **fi)["team"]
    assert ops.gls_describe_config(150, 8, penalty_bits=-2, **fi)["team"] and not ops.gls_describe_config(160, 200, **fi)["team"]   # 8-wave LDS-penalty store
    assert not ops.gls_describe_config(100, 128, penalty_bits=16, **fi)["team"] and not ops.gls_describe_config(300, 8, **fi)["team"]
    # the launch-accurate query follows first_improvement where the older three answer for a best-improvement run
    assert ops.gls_describe_run(30, 1000)["threads"] == 64 and ops.gls_describe_run(30, 1000, first_improvement=True)["threads"] == 128
    assert ops.gls_resident_capacity(100) == 1024 and ops.gls_resident_capacity(50) == 2048
    assert ops.gls_describe_config(50, 1024)["per_cu"] >= 4 and ops.gls_describe_config(50, 2048)["per_cu"] == 8
    assert ops.gls_describe_config(20, 1000)["threads"] == 64 and ops.gls_describe_config(20, 1000)["per_cu"] >= 4
    # single-wavefront workgroups up to n = 33 where the half-wave descent scans exist (32-bit counters), two wavefronts above
    assert ops.gls_describe_config(30, 1000)["threads"] == 64 and ops.gls_describe_config(33, 1000)["threads"] == 64
    assert ops.gls_describe_config(34, 1000)["threads"] == 128 and ops.gls_describe_config(30, 1000, penalty_bits=16)["threads"] == 128
    assert ops.gls_describe_config(100, 512)["store"] == "lds-tri-i32" and ops.gls_describe_config(100, 512)["per_cu"] == 2
    assert ops.gls_describe_config(100, 513)["store"] == "compact"            # 16-bit LDS counters only on request
    assert ops.gls_describe_config(100, 700, penalty_bits=16) == {"store": "lds-tri-u16", "threads": 512, "lds_bytes": 51776, "per_cu": 3, "team": False, "waves_per_simd": 6}
    assert ops.gls_describe_config(200, 256)["store"] == "compact" and ops.gls_describe_config(200, 256)["per_cu"] == 1
    assert ops.gls_describe_config(300, 8)["store"] == "global"
    assert 4 * ops.gls_describe_config(100, 1024)["lds_bytes"] == 160 * 1024


def test_baseline_shapes_run_on_the_scratch_free_instantiations(lib):
    """Every BASELINE.json shape (and the rounds a larger test set is cut into) selects a scratch-free build of the search
    kernel -- the 128-VGPR one (4 resident wavefronts per SIMD; profiles/r03_kernel_resources.txt), or for single-wavefront
    workgroups that fit two per SIMD (TSP20 x 1000) the 256-VGPR one -- and stays fully resident; the 80- / 64-VGPR builds
    (scratch) are only chosen by batches of small instances beyond 16 workgroups per CU."""
    from gnngls_amd import ops
    for n, B in ((20, 1000), (50, 128), (50, 2048), (100, 1024), (100, 625), (100, 1250), (200, 256)):
        c = ops.gls_describe_config(n, B)
        assert c["waves_per_simd"] == (2 if n == 20 else 4), (n, B, c)
        assert c["per_cu"] * 256 >= min(B, ops.gls_resident_capacity(n)), (n, B, c)
    # the 256-VGPR build: single-wavefront workgroups (n = 8 .. 33), at most 2 per SIMD = 2048 instances
    assert ops.gls_describe_config(30, 2048)["waves_per_simd"] == 2 and ops.gls_describe_config(30, 2049)["waves_per_simd"] == 4
    assert ops.gls_describe_config(7, 100)["waves_per_simd"] == 4 and ops.gls_describe_config(34, 100)["waves_per_simd"] == 4
    spilling = [(n, B) for n in (10, 20, 30, 50, 100, 150, 200) for B in (64, 1024, 1536, 3000, 5000, 8192)
                if ops.gls_describe_config(n, B)["waves_per_simd"] not in (2, 4)]
    assert spilling and all(n <= 50 and B > 1024 for n, B in spilling), spilling
    assert ops.gls_describe_config(20, 5000)["waves_per_simd"] in (6, 8)
