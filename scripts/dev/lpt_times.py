#!/usr/bin/env python3
"""Developer tool: kernel time of small / medium batches in three dispatch orders (hmpc_set_dispatch_order):
   natural   -- mode 0: instance b in workgroup b
   predicted -- mode 2: ordered by the cost predicted from the records alone (what a COLD handle gets)
   next tick -- mode 1: ordered by the iteration counts of the solve of the same instances one 5 ms tick earlier
Every timed solve is ONE launch sequence of a batch the handle has not solved before (no order comes from the answer to the same
data).  -> profiles/r05/dispatch_order.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

CASES = [("standing", 10, 1024, 2), ("standing", 10, 2048, 2), ("standing", 10, 8192, 2), ("walking", 10, 1024, 2), ("walking", 10, 8192, 2),
         ("single", 20, 4096, 2), ("standing", 10, 2048, 3), ("standing", 10, 8192, 3), ("standing", 20, 2048, 2)]
print(f"{'case':38s} {'natural':>22s} {'predicted (cold handle)':>32s} {'previous tick':>32s}")
for gait, h, nb, nc in CASES:
    if nc == 3:
        f = synthetic.make_batch3(nb, h, gait, seed=5, hand="contact")
    else:
        f = synthetic.make_batch(nb, h, gait, seed=2, phase="random")
    rec = records.pack_records(f, h, nc)
    rec_next = records.pack_records(synthetic.advance_tick(f, h, seed=9), h, nc)  # the same instances one 5 ms tick later
    t = {}
    for name, mode in (("natural", 0), ("predicted", 2), ("next_tick", 1)):
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb, contacts=nc)
        m.set_dispatch_order(mode)
        ts = []
        for _ in range(5):
            m.upload(rec)
            m.solve()              # tick k (leaves the iteration counts mode 1 sorts by)
            m.upload(rec_next)
            ts.append(m.time_solve(1))   # ONE solve of tick k + 1
        t[name] = min(ts)
        m.close()
    n = t["natural"]
    print(f"{gait:9s} h={h:2d} contacts={nc} b{nb:5d}      {n:7.4f} ms {nb / n / 1e3:7.3f} M/s     "
          f"{t['predicted']:7.4f} ms {nb / t['predicted'] / 1e3:7.3f} M/s {100 * (n / t['predicted'] - 1):+5.1f} %     "
          f"{t['next_tick']:7.4f} ms {nb / t['next_tick'] / 1e3:7.3f} M/s {100 * (n / t['next_tick'] - 1):+5.1f} %", flush=True)
