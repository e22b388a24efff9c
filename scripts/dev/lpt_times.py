#!/usr/bin/env python3
"""Developer tool: kernel time of small / medium batches in natural order and with hmpc_set_dispatch_order (longest previous solve first)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

CASES = [("standing", 10, 1024, 2), ("standing", 10, 2048, 2), ("standing", 10, 8192, 2), ("walking", 10, 1024, 2), ("walking", 10, 8192, 2),
         ("single", 20, 4096, 2), ("standing", 10, 2048, 3), ("standing", 10, 8192, 3), ("standing", 20, 2048, 2)]
for gait, h, nb, nc in CASES:
    if nc == 3:
        f = synthetic.make_batch3(nb, h, gait, seed=5, hand="contact")
    else:
        f = synthetic.make_batch(nb, h, gait, seed=2, phase="random")
    rec = records.pack_records(f, h, nc)
    rec_next = records.pack_records(synthetic.advance_tick(f, h, seed=9), h, nc)  # the same instances one 5 ms tick later
    row = []
    for mode in (False, True):
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb, contacts=nc)
        m.set_dispatch_order(mode)
        m.upload(rec)
        m.solve()
        m.solve()
        row.append(min(m.time_solve(10) for _ in range(3)))
        # the hint one tick old: solve tick k, then time ONE solve of tick k+1 (ordered by tick k's iterations)
        t = []
        for _ in range(4):
            m.upload(rec)
            m.solve()
            m.upload(rec_next)
            t.append(m.time_solve(1))
        row.append(min(t))
        m.close()
    print(f"   next tick: natural {row[1]:7.4f} ms   ordered by the previous tick {row[3]:7.4f} ms  {100 * (row[1] / row[3] - 1):+5.1f} %   ", end="")
    row = [row[0], row[2]]
    print(f"| same batch: {gait:9s} h={h:2d} contacts={nc} b{nb:5d}: natural {row[0]:7.4f} ms {nb / row[0] / 1e3:7.3f} M/s   longest first {row[1]:7.4f} ms {nb / row[1] / 1e3:7.3f} M/s   {100 * (row[0] / row[1] - 1):+5.1f} %")
