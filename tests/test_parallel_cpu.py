"""world_size-2 gloo test (CPU) of the multi-GPU layout: contiguous instance shards, no data-path
collective, one gather of per-instance results to rank 0."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gnngls_amd import parallel
    lo, hi = parallel.shard_range(total, world, rank)
    # "results" of this rank's instances: column 0 = global instance id, column 1 = a cost
    local = torch.stack([torch.arange(lo, hi, dtype=torch.float64), torch.arange(lo, hi, dtype=torch.float64) * 0.5 + 1], 1)
    calls = []
    real_gather, real_all_gather = dist.gather, dist.all_gather
    dist.gather = lambda *a, **k: (calls.append("gather"), real_gather(*a, **k))[1]
    dist.all_gather = lambda *a, **k: (calls.append("all_gather"), real_all_gather(*a, **k))[1]
    out = parallel.gather_results(local, parallel.shard_sizes(total, world))
    dist.gather, dist.all_gather = real_gather, real_all_gather
    assert calls == ["gather"]                     # exactly one collective, no size exchange
    if rank == 0:
        q.put(out.tolist())
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    from gnngls_amd import parallel
    for total in (0, 1, 7, 1024, 10000):
        for world in (1, 2, 3, 8):
            blocks = [parallel.shard_range(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(hi - lo for lo, hi in blocks) <= (total + world - 1) // world


def test_gather_results_gloo_world2_uneven():
    world, total = 2, 7            # 4 + 3 instances: uneven shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert out == [[float(i), i * 0.5 + 1] for i in range(total)]


def test_gather_results_single_process_is_identity():
    from gnngls_amd import parallel
    x = torch.arange(6.).reshape(3, 2)
    assert parallel.gather_results(x) is x


def _cli_worker(rank, world, port, total, q):
    """scripts/test.py's exchange: fixed-width record arrays through ONE tensor gather (no gather_object)."""
    import importlib.util
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gnngls_cli_gather", os.path.join(root, "scripts", "test.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    from gnngls_amd import parallel
    names = [f"i{k}.pkl" for k in range(total)]
    lo, hi = parallel.shard_range(total, world, rank)
    records = []
    for k in range(lo, hi):                                    # instance k: a start row and k % 3 + 1 progress rows
        records.append({"instance": names[k], "time": 100.0 + k, "opt_cost": 2.0 + k})
        records += [{"instance": names[k], "opt_cost": 2.0 + k, "time": 100.0 + k + 0.25 * (r + 1), "cost": 9.0 - r + k}
                    for r in range(k % 3 + 1)]
    calls = []
    real = {name: getattr(dist, name) for name in ("gather", "gather_object", "all_gather", "all_gather_object")}
    for name, f in real.items():
        setattr(dist, name, lambda *a, _f=f, _n=name, **kw: (calls.append(_n), _f(*a, **kw))[1])
    out = cli.gather_records(records, world, rank, names, hi - lo, 4)
    for name, f in real.items():
        setattr(dist, name, f)
    assert calls == ["gather"]
    if rank == 0:
        q.put(out)
    else:
        assert out is None


def test_cli_records_gather_gloo_world2_uneven():
    world, total = 2, 5            # 3 + 2 instances
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    expect = []
    for k in range(total):
        expect.append({"instance": f"i{k}.pkl", "time": 100.0 + k, "opt_cost": 2.0 + k})
        expect += [{"instance": f"i{k}.pkl", "opt_cost": 2.0 + k, "time": 100.0 + k + 0.25 * (r + 1), "cost": 9.0 - r + k}
                   for r in range(k % 3 + 1)]
    assert out == expect
