#!/usr/bin/env python3
"""Developer tool: kernel time vs batch size for the metric's 2-contact case -- the per-instance dependency chain that bounds
small batches (one round of workgroups up to 768 instances) -- in natural dispatch order, ordered by the record-only cost
predictor (a cold handle), and ordered by the previous solve of the same batch.  20 back-to-back launches, best of 3."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

print(f"{'batch':>6s} {'natural':>22s} {'predicted order':>24s} {'previous-solve order':>24s}   active-set iterations of the batch")
for nb in (64, 256, 512, 768, 1024, 1536, 2304):
    f = synthetic.make_batch(nb, 10, "standing", seed=2, phase="random")
    rec = records.pack_records(f, 10)
    row = []
    for mode in (0, 2, 1):
        m = interface.BatchedMPC(synthetic.DT_MPC, 10, synthetic.F_MAX, nb)
        m.set_dispatch_order(mode)
        m.upload(rec)
        m.solve()
        _, st = m.download()
        itmax, itmean = int(interface.status_iters(st).max()), float(interface.status_iters(st).mean())
        row.append(min(m.time_solve(20) for _ in range(3)))
        m.close()
    # (up to 768 instances are ONE round of workgroups: the launch lasts as long as its slowest instance's dependency chain, i.e. it
    #  follows the batch's LARGEST iteration count -- every batch size is its own random draw -- not the batch size)
    print(f"b{nb:5d} " + "   ".join(f"{ms:.4f} ms {nb / ms / 1e3:6.3f} M/s" for ms in row) + f"   iterations mean {itmean:.2f} max {itmax}", flush=True)
