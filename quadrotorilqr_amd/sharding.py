"""Batch sharding across GPUs.  Problems are independent (nothing in ilqr.hh couples two
trajectories), so a batch splits into contiguous shards with no data-path collective; the only
exchange is one gather of the converged trajectories and per-problem scalars to rank 0
(RCCL over xGMI when the backend is "nccl", which is RCCL on ROCm; gloo in the CPU tests).

The gather is ragged and copy-free on the root: every other rank sends its shard point to point and the
root receives it directly into that shard's rows of the result buffer (no padding to the largest shard, no
concatenation afterwards).  xGMI is point to point (7 links per GPU), so seven simultaneous receives on the
root use seven different links; a ring all-gather would put eight times the bytes on every link for a
result only the root wants."""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous shard [lo, hi) of a batch of B problems for `rank` of `world` (sizes differ by <= 1)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(B, world):
    return [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]


def shard_of_step(rank, step, world):
    """Which shard of the global batch `rank` solves at `step` when a sequence of batches is solved: the
    assignment rotates, so that over any `world` consecutive steps every rank has solved every shard once.
    A shard's solve time is set by its slowest problem (data-dependent iteration counts); with a fixed
    assignment the same rank would be the straggler of every step."""
    return (rank + step) % world


import weakref

_warmed_groups = weakref.WeakSet()  # process groups whose communicator exists (the objects themselves: an id() can be reused)
_WORLD_ATTR = "_qilqr_rendezvous_done"


def _first_call_rendezvous(t, group):
    """Once per process group: one tiny all-reduce that every rank enters.  The NCCL (RCCL) communicator of a group is
    created lazily by the first operation on it and every rank must take part in that creation; a ragged gather in
    which some shard is empty (B < world) is entered by the root and the non-empty ranks only, so without this the
    first such call could wait for ranks that never come.  Keyed on the group OBJECT (the default group: on the object
    torch.distributed holds for it, so destroy_process_group + a new init_process_group starts over)."""
    if group is not None:
        g = group
    else:
        try:
            g = dist.distributed_c10d._get_default_group()  # (private: the object torch holds for the default group)
        except Exception:
            g = dist.group.WORLD
    try:
        if g in _warmed_groups:
            return
    except TypeError:  # (a group object that cannot be weakly referenced: fall back to an attribute on it)
        if getattr(g, _WORLD_ATTR, False):
            return
    dist.all_reduce(torch.zeros(1, dtype=torch.float32, device=t.device), group=group)
    try:
        _warmed_groups.add(g)
    except TypeError:
        try:
            setattr(g, _WORLD_ATTR, True)
        except Exception:
            pass  # (the rendezvous is then repeated on every call: correct, one small all-reduce slower)


def gather_to_root(t, sizes, dst=0, group=None, out=None, shard_of_rank=None, rehearse_self=False):
    """Gather per-rank tensors on rank `dst` (a rank of `group`) in SHARD order.

    t              this rank's shard: first dimension = the size of the shard it holds
    sizes          size of every shard, in shard order
    shard_of_rank  which shard each rank holds (default: rank r holds shard r); with a rotating assignment
                   (shard_of_step) pass [shard_of_step(r, step, world) for r in range(world)], so that the
                   result is in global problem order whatever rank solved which shard
    out            optional preallocated result on the root (sum(sizes) rows), reused between calls
    rehearse_self  world size 1 only: instead of returning t, send it to this rank itself through the backend (see below)

    Returns the gathered tensor on the root (rows of shard k at offset sum(sizes[:k])) and None elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        if rehearse_self and dist.is_initialized():
            # A rehearsal of the N > 1 path where only one GPU exists (bench.py --rehearse-nccl, tests/test_gpu_bench_contract.py):
            # the group's communicator is created (first-call all-reduce) and the shard travels through ONE batched
            # send / receive pair addressed to this rank itself -- under the nccl backend ncclSend + ncclRecv in one group on
            # torch's RCCL, in the process that also runs the solver's HIP library -- into the rows `out` reserves for it.
            _first_call_rendezvous(t, group)
            t = t.contiguous()
            if out is None:
                out = torch.empty_like(t)
            if dist.get_backend(group) == "gloo":  # (gloo has no pair from a rank to itself: the CPU tests get the rendezvous and a copy)
                out[:t.shape[0]].copy_(t)
                return out
            me = dist.get_rank(group) if group is None else dist.get_global_rank(group, dist.get_rank(group))
            for wk in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, me, group), dist.P2POp(dist.irecv, out[:t.shape[0]], me, group)]):
                wk.wait()
            return out
        return t
    rank = dist.get_rank(group)
    # before anything that can raise on ONE rank: a rank that leaves here with an exception must not leave the others inside the
    # all-reduce
    _first_call_rendezvous(t, group)
    if shard_of_rank is None:
        shard_of_rank = list(range(world))
    if sorted(shard_of_rank) != list(range(world)):
        raise ValueError("shard_of_rank must be a permutation of the ranks")
    offs = [0]
    for s in sizes:
        offs.append(offs[-1] + int(s))
    mine = shard_of_rank[rank]
    if t.shape[0] != sizes[mine]:
        raise ValueError(f"rank {rank} holds shard {mine} of {sizes[mine]} rows but passed {t.shape[0]}")
    t = t.contiguous()
    # ranks are ranks of `group` throughout; the point-to-point operations address their peers by global rank
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    if rank != dst:
        if t.shape[0] > 0:
            for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, peer(dst), group)]):
                w.wait()
        return None
    if out is None:
        out = torch.empty((offs[-1],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    elif out.shape[0] != offs[-1] or tuple(out.shape[1:]) != tuple(t.shape[1:]) or not out.is_contiguous():
        raise ValueError("out has the wrong shape")
    ops = []
    for r in range(world):
        k = shard_of_rank[r]
        if r == dst or sizes[k] == 0:
            continue
        ops.append(dist.P2POp(dist.irecv, out[offs[k]:offs[k + 1]], peer(r), group))  # a contiguous row range: received in place
    works = dist.batch_isend_irecv(ops) if ops else []
    out[offs[mine]:offs[mine + 1]].copy_(t)
    for w in works:
        w.wait()
    return out
