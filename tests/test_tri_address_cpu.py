"""The select-free packed-triangle address the lean descent scans use (gls_kernels.hip: tri_addr_max): for nodes x != y
the entry D[max(x,y), min(x,y)] of the packed lower triangle sits at index r(hi) + lo with r(v) = v (v - 1) / 2, and
    max(r(x) + y, r(y) + x) == r(max(x,y)) + min(x,y)        whenever x + y >= 3.
The two exceptions are the pairs {0,1} and {0,2}; the kernel keeps node 0 out of the fast path (a lane that owns node 0
passes a large negative row address, a step whose wave-uniform node is 0 takes the exact index).  This test pins the
identity and exactly that exception set, for every n the LDS stores accept."""
import numpy as np


def r(v):
    return v * (v - 1) // 2


def test_identity_and_exceptions():
    n = 256
    x, y = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    off = x != y
    fast = np.maximum(r(x) + y, r(y) + x)
    exact = r(np.maximum(x, y)) + np.minimum(x, y)
    bad = off & (fast != exact)
    pairs = {tuple(sorted(p)) for p in zip(x[bad].tolist(), y[bad].tolist())}
    assert pairs == {(0, 1), (0, 2)}
    assert not (off & (x + y >= 3) & (fast != exact)).any()


def test_no_row_sentinel_wins_for_node_zero():
    """kNoRow = -2^30 as the row address of node 0: the other candidate r(d) + 0 always wins, for any LDS base and node."""
    k_no_row = -(1 << 30)
    for base in (0, 224, 40960, 163840):
        for d in range(1, 256):
            cand_own = k_no_row + 8 * d                    # rx + 8 y with x = node 0
            cand_other = base + 8 * r(d) + 8 * 0           # ry + 8 x
            assert max(cand_own, cand_other) == base + 8 * r(d)


def test_scaled_row_address_has_no_rounding():
    """8 * (e (e - 1) / 2) == 4 e (e - 1): the kernel computes the byte offset of a row without the shift pair."""
    e = np.arange(0, 256)
    assert (8 * r(e) == 4 * e * (e - 1)).all()
