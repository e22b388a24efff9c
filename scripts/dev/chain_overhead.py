#!/usr/bin/env python3
"""Developer tool: what the device-side repair chain costs a NOMINAL solve (nothing flagged: its launches are empty) -- kernel time of
hmpc_time_solve(20) with hmpc_set_device_repair off / on, alternating, three shapes.    python scripts/dev/chain_overhead.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

for name, gait, h, nb in (("standing b8192", "standing", 10, 8192), ("single h20 b4096", "single", 20, 4096), ("mixed b8192", "mixed", 10, 8192), ("standing b1024", "standing", 10, 1024)):
    rec = records.pack_records(synthetic.make_batch(nb, h, gait, seed=2, phase="random"), h)
    ms = {0: [], 1: []}
    hs = {}
    for mode in (0, 1):
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
        m.set_auto_resolve(False)
        m.set_device_repair(mode)
        m.upload(rec)
        m.solve()
        m.time_solve(5)
        hs[mode] = m
    for rep in range(7):
        for mode in (0, 1):
            ms[mode].append(hs[mode].time_solve(20))
    for m in hs.values():
        m.close()
    a, b = np.median(ms[0]), np.median(ms[1])
    print(f"{name:18s} repair off {a:.4f} ms  on {b:.4f} ms  -> {1e3 * (b - a):+.1f} us ({100 * (b - a) / a:+.2f} %)   [min {min(ms[0]):.4f} / {min(ms[1]):.4f}]")
