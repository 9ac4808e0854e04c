"""Batch sharding across GPUs.  Problems are independent (nothing in ilqr.hh couples two
trajectories), so a batch splits into contiguous shards with no data-path collective; the only
exchange is one gather of the converged trajectories and per-problem scalars to rank 0
(RCCL over xGMI when the backend is "nccl", which is RCCL on ROCm; gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous shard [lo, hi) of a batch of B problems for `rank` of `world` (sizes differ by <= 1)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_of_step(rank, step, world):
    """Which shard of the global batch `rank` solves at `step` when a sequence of batches is solved: the
    assignment rotates, so that over any `world` consecutive steps every rank has solved every shard once.
    A shard's solve time is set by its slowest problem (data-dependent iteration counts); with a fixed
    assignment the same rank would be the straggler of every step."""
    return (rank + step) % world


def gather_to_root(t, shard_sizes, dst=0, group=None):
    """Gather per-rank tensors (first dim = that rank's shard size) on rank `dst`; returns the
    concatenation there and None elsewhere.  Ragged shards are padded to the largest one."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return t
    rank = dist.get_rank(group)
    mx = max(shard_sizes)
    if t.shape[0] < mx:
        pad = torch.zeros((mx - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], 0)
    t = t.contiguous()
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
    dist.gather(t, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, shard_sizes)], 0)
