"""The five `torch.ops.gnngls.*` custom operators exist, have shape functions, and have NO CPU kernel."""
import pytest
import torch


@pytest.fixture(scope="module")
def reg():
    from gnngls_amd import torch_ops
    return torch_ops


def test_ops_are_registered_hip_only(reg):
    for name in ("regret_forward", "two_opt_delta_all", "relocate_delta_all", "local_search", "gls_run"):
        assert hasattr(torch.ops.gnngls, name)
    t, D = torch.zeros((1, 6), dtype=torch.int32), torch.zeros((1, 5, 5), dtype=torch.float64)
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.gnngls.two_opt_delta_all(t, D)
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.gnngls.gls_run(D, D[None], t, torch.zeros(1, dtype=torch.float64), 20, 1, 0.0, False, 0)
    with pytest.raises(NotImplementedError, match="CPU"):
        torch.ops.gnngls.regret_forward(torch.zeros((1, 10)), torch.zeros(8), 5, 8, 16, 512, 8)


def test_shape_functions(reg):
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        t, D = torch.empty((3, 11), dtype=torch.int32), torch.empty((3, 10, 10), dtype=torch.float64)
        c = torch.empty(3, dtype=torch.float64)
        assert torch.ops.gnngls.relocate_delta_all(t, D).shape == (3, 11, 11)
        tour, cost, moves = torch.ops.gnngls.local_search(t, c, D, False)
        assert tour.shape == (3, 11) and cost.dtype == torch.float64 and moves.dtype == torch.int32
        out = torch.ops.gnngls.gls_run(D, torch.empty((2, 3, 10, 10), dtype=torch.float64), t, c, 20, 5, 0.0, False, 64)
        assert [tuple(x.shape) for x in out] == [(3, 11), (3,), (3,), (3, 64), (3,)]
        assert torch.ops.gnngls.regret_forward(torch.empty((3, 45)), torch.empty(10), 10, 8, 16, 512, 8).shape == (3, 45)
