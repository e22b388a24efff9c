"""N > 1 path on CPU: world_size-2 gloo processes, contiguous shards, all_gather of the forces.  The per-shard solver
here is the CPU oracle (stand-in: no GPU on this box) -- what is under test is the sharding + gather composition that
bench.py uses unchanged on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hector_simulation_amd import records, sharding, synthetic


def test_shard_bounds_cover_batch():
    for gb in (0, 1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(gb, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, gb, path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_py

    rec = np.load(path)
    lo, hi = sharding.shard_bounds(gb, world, rank)
    sol = oracle_py.solve_records(rec, 10, synthetic.DT_MPC, synthetic.F_MAX, first=lo, count=hi - lo)
    local = torch.from_numpy(sol["q_soln"].astype(np.float32))
    full = sharding.gather_forces(local, gb)
    if rank == 0:
        np.save(path + ".out.npy", full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gb", [8, 7])
def test_two_rank_gather_matches_single_process(oracle, tmp_path, gb):
    f = synthetic.make_batch(gb, 10, "walking", seed=17, phase="random")
    rec = records.pack_records(f, 10)
    path = str(tmp_path / "rec.npy")
    np.save(path, rec)
    mp.spawn(_worker, args=(2, _free_port(), gb, path), nprocs=2, join=True)
    got = np.load(path + ".out.npy")
    want = oracle.solve_records(rec, 10, synthetic.DT_MPC, synthetic.F_MAX)["q_soln"].astype(np.float32)
    np.testing.assert_array_equal(got, want)


def _xch_worker(rank, world, port, shard, path, width=12):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    allf = torch.from_numpy(np.load(path))            # [steps, world*shard, width * h]
    xch = sharding.WrenchExchange(shard, width, "cpu")
    got = []
    for k in range(allf.shape[0]):                    # pipelined exactly as bench.py posts it
        mine = allf[k, rank * shard:(rank + 1) * shard]
        status = (torch.arange(shard, dtype=torch.int32) + 1000 * rank + 100000 * k) | (1 << 30)
        xch.post(k & 1, mine, status)
        if k >= 1:
            w, s = xch.result((k - 1) & 1)
            got.append((w.clone(), s.clone()))
    xch.wait_all()
    w, s = xch.result((allf.shape[0] - 1) & 1)
    got.append((w.clone(), s.clone()))
    if rank == 1:
        torch.save(got, path + ".xch.pt")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("width", [12, 18])
def test_wrench_exchange_pipelined_two_ranks(tmp_path, width):
    """The exchange bench.py runs for N > 1: step-0 wrench + status word of every instance on every rank, double
    buffered (post k+1 while k is in flight).  width 12 = two feet; 18 = BASELINE config 5's three contacts."""
    steps, world, shard = 4, 2, 5
    rng = np.random.default_rng(3)
    allf = rng.standard_normal((steps, world * shard, 10 * width)).astype(np.float32)
    path = str(tmp_path / "f.npy")
    np.save(path, allf)
    mp.spawn(_xch_worker, args=(world, _free_port(), shard, path, width), nprocs=world, join=True)
    got = torch.load(path + ".xch.pt")
    assert len(got) == steps
    for k, (w, s) in enumerate(got):
        np.testing.assert_array_equal(w.numpy(), allf[k, :, :width])
        want = np.concatenate([(np.arange(shard) + 1000 * r + 100000 * k) | (1 << 30) for r in range(world)])
        np.testing.assert_array_equal(s.numpy(), want.astype(np.int32))
