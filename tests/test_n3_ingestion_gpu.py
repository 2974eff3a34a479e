"""SURVEY 8(f) N3 end to end on the GPU: scripts/test.py (the reference's CLI, test.py:20-123) on a dataset directory in the
reference's on-disk formats -- networkx-2.8 instance pickles, scikit-learn-1.0.2 scalers in either layout, a DGL-0.6.1-shaped
checkpoint with Adam's state beside the weights (tests/golden/n3_tsp12/, tests/test_n3_ingestion_cpu.py)."""
import json
import os
import pickle
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "n3_tsp12")
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("layout", ["flat", "edges"])
def test_cli_on_reference_format_files(tmp_path, layout):
    from test_n3_ingestion_cpu import dgl061_checkpoint, names
    data = tmp_path / "tsp12"
    shutil.copytree(FIX, data)
    if layout == "edges":                                                    # datasets.py:48-49 "backward compatability"
        shutil.copy(data / "scalers_edges_layout.pkl", data / "scalers.pkl")
    mdir = tmp_path / "models" / "tsp12"
    mdir.mkdir(parents=True)
    torch.save(dgl061_checkpoint(), mdir / "checkpoint_best_val.pt")         # train.py:59-66
    json.dump({"embed_dim": 128, "n_layers": 3, "n_heads": 8}, open(mdir / "params.json", "w"))     # test.py:31-35
    run_dir = tmp_path / "runs"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "test.py"), str(data / "test.txt"),
                           str(mdir / "checkpoint_best_val.pt"), str(run_dir), "regret_pred", "--time_limit", "0.3",
                           "--use_gpu"], cwd=ROOT)
    df = pickle.load(open(next(run_dir.glob("*.pkl")), "rb"))
    assert set(["instance", "time", "opt_cost", "cost", "best_cost", "gap", "dt"]) <= set(df.columns)
    assert sorted(df["instance"].unique()) == sorted(names())
    last = df.groupby("instance")["gap"].last()
    # `in_solution` of the fixture marks the exact optimum (Held-Karp): no search result is below it, and 0.3 s of guided
    # local search closes TSP12 (test.py:104)
    assert (last > -1e-9).all() and (last < 1e-6).all()
    assert np.allclose(df.groupby("instance")["opt_cost"].first().to_numpy(), df.groupby("instance")["best_cost"].last().to_numpy(), rtol=1e-12)
