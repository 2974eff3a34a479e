"""Multi-GPU layout of the hot path: instances are independent (the reference's loop test.py:59-109
carries no state across instances), so a batch is sharded into contiguous blocks, one process per
GPU, with NO collective on the data path.  The only exchange is one gather of the per-instance
results to rank 0 (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous block [lo, hi) of ceil(total/world) instances for `rank` (last ranks may be short)."""
    per = (total + world - 1) // world
    lo = min(rank * per, total)
    return lo, min(lo + per, total)


def shard_sizes(total, world):
    """Rows every rank contributes: a pure function of (total, world), known everywhere without communication."""
    return [hi - lo for lo, hi in (shard_range(total, world, r) for r in range(world))]


def gather_results(local, sizes=None, dst=0):
    """local [B_local, C] tensor on every rank -> concatenated [sum B_local, C] on rank `dst`, None elsewhere: ONE
    collective (`gather`).  `sizes` = rows per rank (shard_sizes(total, world)) when the shards are uneven; None means
    every rank holds the same number of rows.  No size exchange: the shard layout is a function of (total, world)."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if sizes is None:
        sizes = [local.shape[0]] * world
    sizes = [int(x) for x in sizes]
    if len(sizes) != world or sizes[rank] != local.shape[0]:
        raise ValueError(f"rank {rank}: {local.shape[0]} local rows but the shard layout says {sizes}")
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])
