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
