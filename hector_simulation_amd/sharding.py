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


def _host_staged(dist, group, tensor) -> bool:
    """gloo moves host memory: a device tensor is staged through the host (the CPU tests hand in host tensors; bench.py's
    ``--backend gloo`` test transport hands in device tensors, possibly of ranks that share one GPU)."""
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather_into(dist, out, loc, group, async_op: bool = False):
    if _host_staged(dist, group, loc):
        host_out = out.new_empty(out.shape, device="cpu")
        dist.all_gather_into_tensor(host_out, loc.cpu(), group=group)  # .cpu() waits for the producer stream
        out.copy_(host_out)
        return None
    return dist.all_gather_into_tensor(out, loc, group=group, async_op=async_op)


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
        _all_gather_into(dist, out, local.contiguous(), group)
        return out
    mx = max(sizes)  # ragged: pad every block to the largest shard, gather, cut the padding out
    pad = torch.zeros((mx, width), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    staged = _host_staged(dist, group, pad)
    src = pad.cpu() if staged else pad
    buf = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(buf, src, group=group)
    return torch.cat([b[:s] for b, s in zip(buf, sizes)], dim=0).to(local.device)


class WrenchExchange:
    """The path's one exchange step as SURVEY.md section 8(e) defines it: every rank ends up with the step-0 wrench
    (the 6*contacts floats the controller reads through ``get_solution(0..)``, ConvexMPCLocomotion.cpp:428-429) and
    the status word of every instance of the global batch, in instance order.

    One all_gather of a packed [shard, width+1] float32 block per solve (last column = the status word's bits), with
    ``depth`` slots so that the collective of solve k runs on the communicator's stream while solve k+1 computes:
    ``post(slot, forces, status)`` enqueues pack + all_gather without blocking, ``wait(slot)`` orders the caller's
    stream (nccl) / the host (gloo) after it.  Equal shards only (the bench's weak-scaling layout); ragged batches use
    ``gather_forces``."""

    def __init__(self, shard: int, width: int, device, depth: int = 2, group=None, always_collective: bool = False):
        import torch
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.shard, self.width, self.depth = int(shard), int(width), int(depth)
        self.local = [torch.zeros((shard, width + 1), dtype=torch.float32, device=device) for _ in range(depth)]
        self.out = [torch.zeros((self.world * shard, width + 1), dtype=torch.float32, device=device) for _ in range(depth)]
        self.work = [None] * depth
        # a group of one normally short-cuts to a copy; always_collective issues the all_gather anyway (bench.py
        # --force-exchange: the torchrun code path -- communicator stream, async work handles -- on a one-GPU box)
        self.always_collective = bool(always_collective) and dist.is_available() and dist.is_initialized()

    def post(self, slot: int, forces, status):
        """forces [shard, >= width] float32 and status [shard] int32 of this rank, both on ``device``."""
        import torch

        self.wait(slot)  # the slot's buffers are free again only once its previous collective is done
        loc = self.local[slot]
        loc[:, : self.width].copy_(forces[:, : self.width])
        loc.view(torch.int32)[:, self.width].copy_(status.view(torch.int32))
        if self.world == 1 and not self.always_collective:
            self.out[slot].copy_(loc)
            return
        self.work[slot] = _all_gather_into(self.dist, self.out[slot], loc, self.group, async_op=True)

    def wait(self, slot: int):
        w = self.work[slot]
        if w is not None:
            w.wait()
            self.work[slot] = None

    def wait_all(self):
        for s in range(self.depth):
            self.wait(s)

    def result(self, slot: int):
        """(wrench [global, width] float32, status [global] int32) gathered by the slot's last ``post``."""
        import torch

        self.wait(slot)
        o = self.out[slot]
        return o[:, : self.width], o.view(torch.int32)[:, self.width]
