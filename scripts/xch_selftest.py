#!/usr/bin/env python3
"""Developer check: WrenchExchange on the nccl (RCCL) backend, one rank per GPU (RCCL refuses two ranks on one device,
so this needs a node with at least two GPUs)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hector_simulation_amd import sharding  # noqa: E402


def worker(rank, world, port):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    shard = 1024
    xch = sharding.WrenchExchange(shard, 12, dev)
    ok = True
    for k in range(6):
        f = torch.full((shard, 120), float(10 * k + rank), device="cuda")
        st = torch.full((shard,), 7 * k + rank, dtype=torch.int32, device="cuda")
        xch.post(k & 1, f, st)
        if k >= 1:
            w, s = xch.result((k - 1) & 1)
            torch.cuda.synchronize()
            for r in range(world):
                ok &= bool((w[r * shard:(r + 1) * shard] == float(10 * (k - 1) + r)).all())
                ok &= bool((s[r * shard:(r + 1) * shard] == 7 * (k - 1) + r).all())
    xch.wait_all()
    torch.cuda.synchronize()
    print(f"rank {rank}: exchange ok = {ok}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29533), nprocs=2, join=True)  # one rank per GPU: edit worker() device indices accordingly
