"""Multi-GPU layout of a batch of independent MPC instances.

Every instance is an independent QP, so the batch is cut into contiguous shards, one per rank (one process per GPU),
and the data path needs no collective; the single exchange step is the gather of the solved forces
(all_gather over RCCL -- torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests)."""
from __future__ import annotations


def shard_bounds(global_batch: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of rank; the first (global_batch % world) ranks carry one extra instance."""
    if world < 1 or not (0 <= rank < world) or global_batch < 0:
        raise ValueError((global_batch, world, rank))
    base, extra = divmod(global_batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_forces(local, global_batch: int, group=None):
    """all_gather of the per-rank force blocks [shard, 12h] into [global_batch, 12h] in instance order, on every rank."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    width = local.shape[1]
    sizes = [shard_bounds(global_batch, world, r)[1] - shard_bounds(global_batch, world, r)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((global_batch, width), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    mx = max(sizes)  # ragged: pad every block to the largest shard, gather, cut the padding out
    pad = torch.zeros((mx, width), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(buf, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(buf, sizes)], dim=0)
