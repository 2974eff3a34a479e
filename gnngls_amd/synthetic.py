"""Seeded synthetic inputs (all data/ and models/ files of the reference are git-LFS pointer stubs).

Instances follow scripts/generate_instances.py:25-33: uniform points in the unit square, complete
graph, weight = Euclidean distance in fp64; node 0 is the depot."""
import numpy as np


def random_instances(rng, B, n):
    """Returns (D [B,n,n] fp64 symmetric with zero diagonal, pos [B,n,2])."""
    pos = rng.random((B, n, 2))
    d = pos[:, :, None, :] - pos[:, None, :, :]
    D = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1])
    D = np.triu(D, 1)
    D = D + D.transpose(0, 2, 1)
    return D, pos
