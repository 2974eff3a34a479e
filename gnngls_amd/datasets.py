"""Mirror of the input surface of gnngls/datasets.py (reference datasets.py:37-95): `TSPDataset` with
the same constructor, attributes (`instances`, `root_dir`, `scalers`, `G`) and `get_scaled_features`.

Differences forced by the environment, none on the arithmetic:
  * DGL is not needed: `self.G` / the returned graphs are `gnngls_amd.models.LineGraph` objects
    (the line graph of K_n has closed-form structure); `ndata` carries the same keys.
  * networkx >= 3 removed read_gpickle (datasets.py:56,69): instances are read with pickle.load.
  * git-LFS pointer stubs (every file under the reference's data/ and models/) are detected and
    reported instead of failing inside pickle.
Label generation (`set_labels`, needs LKH) is out of scope; `set_features` is kept.
"""
import pathlib
import pickle

import numpy as np
import torch

from .models import LineGraph

LFS_MAGIC = b"version https://git-lfs.github.com/spec/v1"


def is_lfs_pointer(path):
    with open(path, "rb") as f:
        return f.read(len(LFS_MAGIC)) == LFS_MAGIC


def read_gpickle(path):
    """nx.read_gpickle (datasets.py:56,69; gone in networkx 3) = pickle.load.  The reference's files were written by
    networkx 2.8 (Pipfile.lock): such a Graph unpickles under networkx 3.x with the views its generating code touched
    (`nodes`, `edges`, `adj`, `degree`: cached properties) in the instance dict and without the `__networkx_cache__` slot
    3.x creates in `Graph.__init__`; the slot is added here so the object is a complete 3.x graph (fixture:
    tests/golden/n3_tsp12/, tests/test_n3_ingestion_cpu.py)."""
    if is_lfs_pointer(path):
        raise FileNotFoundError(f"{path} is a git-LFS pointer stub, not the real object (fetch it with git lfs pull)")
    with open(path, "rb") as f:
        obj = pickle.load(f)
    if hasattr(obj, "_adj") and hasattr(obj, "_node") and "__networkx_cache__" not in getattr(obj, "__dict__", {}):
        obj.__dict__["__networkx_cache__"] = {}
    return obj


def set_features(G):
    """datasets.py:14-20"""
    for e in G.edges:
        G.edges[e]["features"] = np.array([G.edges[e]["weight"]], dtype=np.float32)


class TSPDataset(torch.utils.data.Dataset):
    def __init__(self, instances_file, scalers_file=None, feat_drop_idx=[]):
        if not isinstance(instances_file, pathlib.Path):
            instances_file = pathlib.Path(instances_file)
        self.root_dir = instances_file.parent
        if is_lfs_pointer(instances_file):
            raise FileNotFoundError(f"{instances_file} is a git-LFS pointer stub")
        self.instances = [line.strip() for line in open(instances_file)]

        if scalers_file is None:
            scalers_file = self.root_dir / "scalers.pkl"
        scalers = read_gpickle(scalers_file)
        if "edges" in scalers:  # for backward compatability (datasets.py:48-49)
            self.scalers = scalers["edges"]
        else:
            self.scalers = scalers

        self.feat_drop_idx = feat_drop_idx

        # only works for homogenous datasets (datasets.py:55-60)
        G = read_gpickle(self.root_dir / self.instances[0])
        self.G = LineGraph(len(G.nodes))

    def __len__(self):
        return len(self.instances)

    def __getitem__(self, i):
        if torch.is_tensor(i):
            i = i.tolist()
        G = read_gpickle(self.root_dir / self.instances[i])
        return self.get_scaled_features(G)

    def get_scaled_features(self, G):
        """datasets.py:73-95 (features / regret gathered in line-graph node order, MinMax-scaled)."""
        es = self.G.ndata["e"].numpy()
        features = np.vstack([G.edges[tuple(e)]["features"] for e in es])
        regret = np.vstack([G.edges[tuple(e)]["regret"] for e in es])
        features_transformed = self.scalers["features"].transform(features)
        features_transformed = np.delete(features_transformed, self.feat_drop_idx, axis=1)
        regret_transformed = self.scalers["regret"].transform(regret)

        H = LineGraph(self.G.n)
        H.ndata["features"] = torch.tensor(features_transformed, dtype=torch.float32)
        H.ndata["regret"] = torch.tensor(regret_transformed, dtype=torch.float32)
        H.ndata["in_solution"] = torch.tensor(regret, dtype=torch.float32)      # sic, datasets.py:94
        return H
