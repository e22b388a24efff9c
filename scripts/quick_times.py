#!/usr/bin/env python3
"""Developer tool: kernel time (HIP events, hmpc_time_solve) of the BASELINE shapes on the current build."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (brings the HIP runtime up first, see tests/conftest.py)

torch.zeros(1, device="cuda")
from hector_simulation_amd import interface, records, synthetic  # noqa: E402

CASES = [("standing_b8192", "standing", 10, 8192, 2), ("standing_b1024", "standing", 10, 1024, 2), ("walking_b8192", "walking", 10, 8192, 2),
         ("walking_b1024", "walking", 10, 1024, 2), ("h20_single_b4096", "single", 20, 4096, 2), ("3contact_b2048", "standing", 10, 2048, 3),
         ("3contact_b8192", "standing", 10, 8192, 3), ("h20_double_b2048", "standing", 20, 2048, 2), ("h14_double_b2048", "standing", 14, 2048, 2)]
only = set(sys.argv[1:])
for name, gait, h, nb, nc in CASES:
    if only and name not in only:
        continue
    if nc == 3:
        f = synthetic.make_batch3(nb, h, gait, seed=5, hand="contact")
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb, contacts=3)
        m.upload(records.pack_records(f, h, 3))
    else:
        f = synthetic.make_batch(nb, h, gait, seed=2, phase="random")
        m = interface.BatchedMPC(synthetic.DT_MPC, h, synthetic.F_MAX, nb)
        m.upload(records.pack_records(f, h))
    m.solve()
    _, st = m.download()
    ms = min(m.time_solve(10) for _ in range(3))
    it = interface.status_iters(st)
    print(f"{name:18s} {ms:8.4f} ms  {nb / ms / 1e3:8.3f} M solves/s  failed {int((interface.status_code(st) != 0).sum())}  iters mean {it.mean():.2f}")
    m.close()
